// lm_head cross-entropy over the 32200-entry vocabulary and the train_step loss reduction.
// Bandwidth-bound: fwd reads R*V f32 logits once (+1 for the exp pass, L2-resident), bwd reads them once and
// writes R*V bf16 dlogits.
#include "common.h"
#include "vlt5_hip.h"

namespace {

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
    v = is_max ? wave_max(v) : wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < nw; ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
    return r;
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                     float* __restrict__ loss_tok, float* __restrict__ lse, int R, int V) {
    // ONE pass over the row: every thread keeps a running (max, sum of exp relative to it) and the partials are merged at the end
    // -- the row (129 KB at V = 32200) is read once instead of twice
    __shared__ float sh[4];
    const int r = blockIdx.x;
    const float* row = logits + (size_t)r * V;
    float m = -INFINITY, s = 0.f;
    // (eight quads requested before the first is consumed, same order of arithmetic: 400 workgroups are 6 waves per CU, and one dependent
    // load per iteration made the 32 iterations of a thread 32 round trips)
    for (int c0 = threadIdx.x * 4; c0 < V; c0 += 8 * 1024) {
        float4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u * 1024;
            q[u] = c < V ? *reinterpret_cast<const float4*>(row + c) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (c0 + u * 1024 < V) {
                const float4 v = q[u];
                const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
                if (mx > m) { s *= fast_exp(m - mx); m = mx; }
                s += (fast_exp(v.x - m) + fast_exp(v.y - m)) + (fast_exp(v.z - m) + fast_exp(v.w - m));
            }
        }
    }
    const float M = block_reduce(m, true, sh);
    s = block_reduce(s * fast_exp(m - M), false, sh);          // (a thread without elements: m = -inf, s = 0 -> 0 * exp(-inf) = 0)
    if (threadIdx.x == 0) {
        const float l = M + logf(s);
        lse[r] = l;
        const long long y = labels[r];
        loss_tok[r] = (y >= 0 && y < V) ? (l - row[y]) : 0.f;          // ignore_index = -100 -> 0
    }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                     const float* __restrict__ lse, const float* __restrict__ row_w,
                                                     const float* __restrict__ gout, bf16_t* __restrict__ dlogits, int R, int V) {
    const int r = blockIdx.x;
    const float* row = logits + (size_t)r * V;
    bf16_t* drow = dlogits + (size_t)r * V;
    const long long y = labels[r];
    const bool valid = (y >= 0 && y < V);
    const float w = valid ? row_w[r] * (gout ? gout[0] : 1.f) : 0.f;
    const float l = lse[r];
    for (int c0 = threadIdx.x * 4; c0 < V; c0 += 8 * 1024) {
        float4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u * 1024;
            q[u] = c < V ? *reinterpret_cast<const float4*>(row + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u * 1024;
            if (c < V) {
                float o[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float p = valid ? fast_exp(o[k] - l) : 0.f;
                    if (c + k == y) p -= 1.f;
                    o[k] = p * w;
                }
                uint2 pk;
                pk.x = pack_bf16x2(o[0], o[1]);
                pk.y = pack_bf16x2(o[2], o[3]);
                *reinterpret_cast<uint2*>(drow + c) = pk;
            }
        }
    }
}

// one block; thread b handles sample b (grid-stride), fixed-order final sum by thread 0
__global__ void loss_reduce_kernel(const float* __restrict__ loss_tok, const long long* __restrict__ labels,
                                   const float* __restrict__ scores, float* __restrict__ loss, float* __restrict__ row_w, int B,
                                   int T) {
    extern __shared__ float per[];
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float cnt = 0.f, s = 0.f;
        for (int t = 0; t < T; ++t) {
            const bool m = labels[b * T + t] != -100;
            cnt += m ? 1.f : 0.f;
            s += m ? loss_tok[b * T + t] : 0.f;
        }
        const float den = fmaxf(cnt, 1.f);
        per[b] = (s / den) * scores[b];
        if (row_w)
            for (int t = 0; t < T; ++t)
                row_w[b * T + t] = (labels[b * T + t] != -100) ? scores[b] / (den * (float)B) : 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int b = 0; b < B; ++b) tot += per[b];
        loss[0] = tot / (float)B;
    }
}

// first index of the row maximum (torch.argmax semantics for ties: the lowest index), one workgroup per row
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int cols, long long* __restrict__ out) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const float* r = x + (size_t)blockIdx.x * cols;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float v = r[c];
        if (v > best || (v == best && c < idx)) { best = v; idx = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        out[blockIdx.x] = idx == 0x7fffffff ? 0 : idx;       // an all-NaN row: index 0
    }
}

}  // namespace

extern "C" int vlt5_ce_fwd(const float* logits, const long long* labels, float* loss_tok, float* lse, int R, int V, void* stream) {
    if (!logits || !labels || !loss_tok || !lse || R <= 0 || V <= 0) return VLT5_ERR_ARG;
    if (V & 3) return VLT5_ERR_ALIGN;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, labels, loss_tok, lse, R, V);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_ce_bwd(const float* logits, const long long* labels, const float* lse, const float* row_w, const float* gout,
                           void* dlogits_bf16, int R, int V, void* stream) {
    if (!logits || !labels || !lse || !row_w || !dlogits_bf16 || R <= 0 || V <= 0) return VLT5_ERR_ARG;
    if (V & 3) return VLT5_ERR_ALIGN;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, labels, lse, row_w, gout,
                       (bf16_t*)dlogits_bf16, R, V);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_loss_reduce(const float* loss_tok, const long long* labels, const float* scores, float* loss, float* row_w,
                                int B, int T, void* stream) {
    if (!loss_tok || !labels || !scores || !loss || B <= 0 || T <= 0) return VLT5_ERR_ARG;
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), B * sizeof(float), (hipStream_t)stream, loss_tok, labels, scores,
                       loss, row_w, B, T);
    LAUNCH_CHECK();
    return VLT5_OK;
}

extern "C" int vlt5_argmax_rows(const float* x, int rows, int cols, long long* out, void* stream) {
    if (!x || !out || rows <= 0 || cols <= 0) return VLT5_ERR_ARG;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, cols, out);
    LAUNCH_CHECK();
    return VLT5_OK;
}
