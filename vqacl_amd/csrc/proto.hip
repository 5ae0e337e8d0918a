// SS/SI prototype head: token mean-pooling, per-class batch means, the per-task EMA state update and the
// cosine-similarity retrieval (integer argmax).  Tiny tensors ([<=80, d]); everything stays in f32 so the
// integer prototype indices match the CPU path whenever the top-2 margin exceeds the encoder tolerance.
#include "common.h"
#include "vlt5_hip.h"

namespace {

__global__ void pool_kernel(const float* __restrict__ hidden, long long sb, int B, int S, int d, int split, float* __restrict__ poolQ,
                            float* __restrict__ poolV) {
    const int b = blockIdx.x;
    const int nq = min(split, S), nv = S - nq;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        const float* h = hidden + (size_t)b * sb + c;
        float sq = 0.f, sv = 0.f;
        for (int s = 0; s < nq; ++s) sq += h[(size_t)s * d];
        for (int s = nq; s < S; ++s) sv += h[(size_t)s * d];
        poolQ[(size_t)b * d + c] = sq / (float)nq;
        poolV[(size_t)b * d + c] = sv / (float)nv;          // nv == 0 -> NaN, as torch.mean of an empty slice
    }
}

// one block per class; the (few) members of the class are listed in LDS first so the column loop only visits them
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
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) {
            const float w = wgt[b];
            if (w != 0.f) s += w * pool[(size_t)b * d + c];
        }
        proto[(size_t)cls * d + c] = s / div;
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
    extern __shared__ float lds[];                         // [d] tanh(x) then [C] similarities
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
    for (int cls = wave; cls < C; cls += 4) {
        const float* a = An + (size_t)cls * d;
        float dot = 0.f;
        for (int c = lane; c < d; c += 64) dot += a[c] * tx[c];
        dot = wave_sum(dot);
        if (lane == 0) sim[cls] = dot / nb;
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
    hipLaunchKernelGGL(pool_kernel, dim3(B), dim3(256), 0, ST, hidden, sb, B, S, d, split, poolQ, poolV);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_proto_class_mean(const float* pool, const float* onehot, float* proto, float* cnt, int B, int C, int d,
                                     void* stream) {
    if (!pool || !onehot || !proto || !cnt || B <= 0 || C <= 0) return VLT5_ERR_ARG;
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
