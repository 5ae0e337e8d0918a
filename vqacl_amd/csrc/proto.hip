// SS/SI prototype head: token mean-pooling, per-class batch means, the per-task EMA state update and the
// cosine-similarity retrieval (integer argmax).  Tiny tensors ([<=80, d]); everything stays in f32 so the
// integer prototype indices match the CPU path whenever the top-2 margin exceeds the encoder tolerance.
#include "common.h"
#include "vlt5_hip.h"

namespace {

// one block of 8 waves per sample: wave w sums tokens w, w+8, ... (independent 16-byte loads, d/256 in flight per lane and
// token), the eight partial sums meet in LDS.  d % 4 == 0, d <= 2048.
__global__ __launch_bounds__(512) void pool_kernel(const float* __restrict__ hidden, long long sb, int B, int S, int d, int split,
                                                   float* __restrict__ poolQ, float* __restrict__ poolV) {
    extern __shared__ __attribute__((aligned(16))) float psum[];   // [2][8][d]
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = min(split, S), nv = S - nq;
    const float* h = hidden + (size_t)b * sb;
    float4 aq[8], av[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { aq[j] = make_float4(0.f, 0.f, 0.f, 0.f); av[j] = aq[j]; }
    for (int s = wave; s < S; s += 8) {
        const bool isq = s < nq;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = lane * 4 + j * 256;
            if (c < d) {
                const float4 v = *reinterpret_cast<const float4*>(h + (size_t)s * d + c);
                if (isq) { aq[j].x += v.x; aq[j].y += v.y; aq[j].z += v.z; aq[j].w += v.w; }
                else     { av[j].x += v.x; av[j].y += v.y; av[j].z += v.z; av[j].w += v.w; }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = lane * 4 + j * 256;
        if (c < d) {
            *reinterpret_cast<float4*>(psum + (size_t)wave * d + c) = aq[j];
            *reinterpret_cast<float4*>(psum + (size_t)(8 + wave) * d + c) = av[j];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 512) {
        float sq = 0.f, sv = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { sq += psum[(size_t)w * d + c]; sv += psum[(size_t)(8 + w) * d + c]; }
        poolQ[(size_t)b * d + c] = sq / (float)nq;
        poolV[(size_t)b * d + c] = sv / (float)nv;          // nv == 0 -> NaN, as torch.mean of an empty slice
    }
}

// one block per class: proto[cls] = onehot[:, cls]^T pool / max(count, 1).  The product is taken over ALL samples (weight 0
// for non-members, exactly the matrix product of the reference), so the loads of the loop are independent and pipeline.
__global__ __launch_bounds__(256) void class_mean_kernel(const float* __restrict__ pool, const float* __restrict__ onehot,
                                                         float* __restrict__ proto, float* __restrict__ cnt, int B, int C, int d) {
    extern __shared__ float wgt[];                 // [B] one-hot column of this class
    const int cls = blockIdx.x;
    for (int b = threadIdx.x; b < B; b += blockDim.x) wgt[b] = onehot[(size_t)b * C + cls];
    __syncthreads();
    float n = 0.f;
    for (int b = 0; b < B; ++b) n += wgt[b];
    if (threadIdx.x == 0) cnt[cls] = n;
    const float div = n <= 0.f ? 1.f : n;
    for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int b = 0; b < B; ++b) {
            const float w = wgt[b];
            const float4 v = *reinterpret_cast<const float4*>(pool + (size_t)b * d + c);
            s.x += w * v.x; s.y += w * v.y; s.z += w * v.z; s.w += w * v.w;
        }
        *reinterpret_cast<float4*>(proto + (size_t)cls * d + c) = make_float4(s.x / div, s.y / div, s.z / div, s.w / div);
    }
}

// update_prototype, elementwise over max(CQ,CV) x d
__global__ void proto_update_kernel(const float* __restrict__ curQ, const float* __restrict__ curV, const float* __restrict__ numQ,
                                    const float* __restrict__ numV, float* __restrict__ Qp, float* __restrict__ Vp,
                                    float* __restrict__ Qnum, float* __restrict__ Vnum, float* __restrict__ qmem, int qmem_init,
                                    int first, int task, float alpha, float beta, int CQ, int CV, int d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < CV * d) Vp[i] = first ? curV[i] : beta * Vp[i] + (1.f - beta) * curV[i];
    if (i < CQ * d) {
        const int row = i / d;
        if (first) {
            if (task == 0 || row == task) Qp[i] = curQ[i];
        } else if (task == 0) {
            Qp[i] = curQ[i];
        } else {
            const float now = (row == task) ? 0.f : curQ[i];
            float mem = qmem_init ? alpha * qmem[i] + (1.f - alpha) * now : now;
            if (row == task) mem = curQ[i];         // the reference writes the current row through an alias of the memory
            qmem[i] = mem;
            Qp[i] = mem;
        }
    }
    if (i < CQ) Qnum[i] = first ? numQ[i] : Qnum[i] + numQ[i];
    if (i < CV) Vnum[i] = first ? numV[i] : Vnum[i] + numV[i];
}

// tanh-normalised prototypes, computed once per call: An[c] = tanh(P_c) / max(||tanh(P_c)||, 1e-12)
__global__ __launch_bounds__(256) void proto_normalize_kernel(const float* __restrict__ protos, float* __restrict__ An, int C, int d) {
    __shared__ float sh[4];
    const int cls = blockIdx.x;
    const float* pr = protos + (size_t)cls * d;
    float s = 0.f;
    for (int c = threadIdx.x; c < d; c += 256) { float t = tanhf(pr[c]); s += t * t; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    const float inv = 1.0f / fmaxf(sqrtf((sh[0] + sh[1]) + (sh[2] + sh[3])), 1e-12f);
    for (int c = threadIdx.x; c < d; c += 256) An[(size_t)cls * d + c] = tanhf(pr[c]) * inv;
}

// one block per sample: cosine similarity against every (pre-normalised) prototype, first-max argmax, gather
__global__ __launch_bounds__(256) void retrieve_kernel(const float* __restrict__ protos, const float* __restrict__ An,
                                                       const float* __restrict__ pool, long long* __restrict__ idx,
                                                       float* __restrict__ out_f32, long long sb, bf16_t* __restrict__ out_bf16,
                                                       long long sb16, int B, int C, int d) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [d] tanh(x) then [C] similarities
    float* tx = lds;
    float* sim = lds + d;
    __shared__ float part[4];
    __shared__ int best_sh;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* x = pool + (size_t)b * d;
    float s = 0.f;
    for (int c = threadIdx.x; c < d; c += 256) { float t = tanhf(x[c]); tx[c] = t; s += t * t; }
    s = wave_sum(s);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    const float nb = fmaxf(sqrtf((part[0] + part[1]) + (part[2] + part[3])), 1e-12f);
    // wave w scores classes w, w+4, ...: four classes per round so 4 x (d/256) independent 16-byte loads are in flight per lane
    for (int cls0 = wave; cls0 < C; cls0 += 16) {
        float dot[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = lane * 4; c < d; c += 256) {
            const float4 t = *reinterpret_cast<const float4*>(tx + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cls = cls0 + 4 * u;
                if (cls < C) {
                    const float4 a = *reinterpret_cast<const float4*>(An + (size_t)cls * d + c);
                    dot[u] += (a.x * t.x + a.y * t.y) + (a.z * t.z + a.w * t.w);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cls = cls0 + 4 * u;
            const float r = wave_sum(dot[u]);
            if (lane == 0 && cls < C) sim[cls] = r / nb;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0;
        float bv = sim[0];
        for (int cls = 1; cls < C; ++cls)
            if (sim[cls] > bv) { bv = sim[cls]; best = cls; }
        best_sh = best;
        idx[b] = best;
    }
    __syncthreads();
    const float* sel = protos + (size_t)best_sh * d;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float v = sel[c];
        if (out_f32) out_f32[b * sb + c] = v;
        if (out_bf16) out_bf16[b * sb16 + c] = f32_to_bf16(v);
    }
}

__global__ __launch_bounds__(256) void memory_loss_kernel(const float* __restrict__ pool, const float* __restrict__ onehot,
                                                          const float* __restrict__ protos, float* __restrict__ out, int B, int C,
                                                          int d) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < B * d; i += blockDim.x) {
        const int b = i / d, c = i % d;
        float t = 0.f;
        for (int cls = 0; cls < C; ++cls) t += onehot[(size_t)b * C + cls] * protos[(size_t)cls * d + c];
        const float diff = pool[i] - t;
        acc += diff * diff;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (sh[0] + sh[1] + sh[2] + sh[3]) / (float)B;
}

}  // namespace

#define ST ((hipStream_t)stream)
extern "C" int vlt5_proto_pool(const float* hidden, long long sb, int B, int S, int d, int split, float* poolQ, float* poolV,
                               void* stream) {
    if (!hidden || !poolQ || !poolV || B <= 0 || S <= 0 || d <= 0 || split <= 0) return VLT5_ERR_ARG;
    if ((d & 3) || d > 2048 || (sb & 3)) return VLT5_ERR_ALIGN;
    static std::atomic<unsigned long long> optin{0};        // 16*d floats of dynamic LDS: above 64 KB for d > 1024
    if (int rc = vlt5_lds_optin(reinterpret_cast<const void*>(&pool_kernel), 16 * 2048 * 4, optin)) return rc;
    hipLaunchKernelGGL(pool_kernel, dim3(B), dim3(512), 16 * (size_t)d * sizeof(float), ST, hidden, sb, B, S, d, split, poolQ, poolV);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_proto_class_mean(const float* pool, const float* onehot, float* proto, float* cnt, int B, int C, int d,
                                     void* stream) {
    if (!pool || !onehot || !proto || !cnt || B <= 0 || C <= 0) return VLT5_ERR_ARG;
    if (d & 3) return VLT5_ERR_ALIGN;
    hipLaunchKernelGGL(class_mean_kernel, dim3(C), dim3(256), B * sizeof(float), ST, pool, onehot, proto, cnt, B, C, d);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_proto_update(const float* curQ, const float* curV, const float* numQ, const float* numV, float* Qproto,
                                 float* Vproto, float* Qnum, float* Vnum, float* qmem, int qmem_initialised, int first, int task,
                                 float alpha, float beta, int CQ, int CV, int d, void* stream) {
    if (!curQ || !curV || !numQ || !numV || !Qproto || !Vproto || !Qnum || !Vnum) return VLT5_ERR_ARG;
    if (task < 0 || task >= CQ) return VLT5_ERR_ARG;
    if (!first && task != 0 && !qmem) return VLT5_ERR_ARG;
    int n = (CQ > CV ? CQ : CV) * d;
    hipLaunchKernelGGL(proto_update_kernel, dim3((n + 255) / 256), dim3(256), 0, ST, curQ, curV, numQ, numV, Qproto, Vproto, Qnum,
                       Vnum, qmem, qmem_initialised, first, task, alpha, beta, CQ, CV, d);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_proto_retrieve(const float* protos, const float* pool, long long* idx, float* out_f32, long long sb,
                                   void* out_bf16, long long sb_bf16, float* scratch, int B, int C, int d, void* stream) {
    if (!protos || !pool || !idx || !scratch || B <= 0 || C <= 0) return VLT5_ERR_ARG;
    if (d & 3) return VLT5_ERR_ALIGN;
    hipLaunchKernelGGL(proto_normalize_kernel, dim3(C), dim3(256), 0, ST, protos, scratch, C, d);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(retrieve_kernel, dim3(B), dim3(256), (d + C) * sizeof(float), ST, protos, scratch, pool, idx, out_f32, sb,
                       (bf16_t*)out_bf16, sb_bf16, B, C, d);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_proto_memory_loss(const float* pool, const float* onehot, const float* protos, float* out, int B, int C, int d,
                                      void* stream) {
    if (!pool || !onehot || !protos || !out) return VLT5_ERR_ARG;
    hipLaunchKernelGGL(memory_loss_kernel, dim3(1), dim3(256), 0, ST, pool, onehot, protos, out, B, C, d);
    LAUNCH_CHECK();
    return VLT5_OK;
}
