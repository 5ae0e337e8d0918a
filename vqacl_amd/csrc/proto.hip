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

// ---- the fused head (vlt5_proto_head_fwd): per class row, everything between the pooled features and the retrieval -----------
// One workgroup per prototype row (CQ question rows, then CV category rows): the class mean of the batch (calculate_current_prototype),
// the per-task state update of that row (update_prototype) and its tanh-normalised copy for the cosine retrieval -- the same
// arithmetic, in the same order, as class_mean_kernel + proto_update_kernel + proto_normalize_kernel (bit-identical results), in
// one launch instead of five.  update == 0 (evaluation, proto_update=False): only the normalised copy.
struct HeadArgs {
    const float *poolQ, *poolV, *onehotQ, *onehotV;
    float *Qp, *Vp, *Qnum, *Vnum, *qmem, *AnQ, *AnV;
    int qmem_init, first, task, update;
    float alpha, beta;
    int B, CQ, CV, d;
    float* packed; int phase;          // data parallel: 1 = class sums + counts of the local batch -> packed; 2 = the update from the all-reduced packed
};
// packed = [sums Q (CQ*d) | counts Q (CQ) | sums V (CV*d) | counts V (CV)]  (the layout of vlt5_proto_stats_pack)
__device__ __forceinline__ float* packed_sums(const HeadArgs& a, bool isQ, int cls) {
    return a.packed + (isQ ? (size_t)cls * a.d : (size_t)a.CQ * (a.d + 1) + (size_t)cls * a.d);
}
__device__ __forceinline__ float* packed_count(const HeadArgs& a, bool isQ, int cls) {
    return a.packed + (isQ ? (size_t)a.CQ * a.d + cls : (size_t)a.CQ * (a.d + 1) + (size_t)a.CV * a.d + cls);
}
__global__ __launch_bounds__(256) void proto_row_kernel(HeadArgs a) {
    extern __shared__ float wgt[];                 // [B] one-hot column of this class
    __shared__ float sh[4];
    const bool isQ = (int)blockIdx.x < a.CQ;
    const int cls = isQ ? blockIdx.x : blockIdx.x - a.CQ, C = isQ ? a.CQ : a.CV, d = a.d;
    const float* pool = isQ ? a.poolQ : a.poolV;
    const float* onehot = isQ ? a.onehotQ : a.onehotV;
    float* P = (isQ ? a.Qp : a.Vp) + (size_t)cls * d;
    float* An = (isQ ? a.AnQ : a.AnV) + (size_t)cls * d;
    if (a.update) {
        float n = 0.f;
        if (a.phase != 2) {
            for (int b = threadIdx.x; b < a.B; b += blockDim.x) wgt[b] = onehot[(size_t)b * C + cls];
            __syncthreads();
            for (int b = 0; b < a.B; ++b) n += wgt[b];
        } else {
            n = *packed_count(a, isQ, cls);            // the global batch's count of this class
        }
        if (a.phase == 1) {                            // local class sums + count, nothing else (the all-reduce comes next)
            if (threadIdx.x == 0) *packed_count(a, isQ, cls) = n;
            float* dst = packed_sums(a, isQ, cls);
            for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) {
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
                for (int b = 0; b < a.B; ++b) {
                    const float w = wgt[b];
                    const float4 v = *reinterpret_cast<const float4*>(pool + (size_t)b * d + c);
                    s.x += w * v.x; s.y += w * v.y; s.z += w * v.z; s.w += w * v.w;
                }
                dst[c] = s.x; dst[c + 1] = s.y; dst[c + 2] = s.z; dst[c + 3] = s.w;
            }
            return;
        }
        const float div = n <= 0.f ? 1.f : n;
        if (threadIdx.x == 0) {
            float* num = isQ ? a.Qnum : a.Vnum;
            num[cls] = a.first ? n : num[cls] + n;
        }
        for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.phase == 2) {
                const float* src = packed_sums(a, isQ, cls);
                s = make_float4(src[c], src[c + 1], src[c + 2], src[c + 3]);
            } else {
#pragma unroll 8
                for (int b = 0; b < a.B; ++b) {
                    const float w = wgt[b];
                    const float4 v = *reinterpret_cast<const float4*>(pool + (size_t)b * d + c);
                    s.x += w * v.x; s.y += w * v.y; s.z += w * v.z; s.w += w * v.w;
                }
            }
            const float cur[4] = {s.x / div, s.y / div, s.z / div, s.w / div};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = cls * d + c + e;
                if (!isQ) {
                    P[c + e] = a.first ? cur[e] : a.beta * P[c + e] + (1.f - a.beta) * cur[e];
                } else if (a.first) {
                    if (a.task == 0 || cls == a.task) P[c + e] = cur[e];
                } else if (a.task == 0) {
                    P[c + e] = cur[e];
                } else {
                    const float now = (cls == a.task) ? 0.f : cur[e];
                    float mem = a.qmem_init ? a.alpha * a.qmem[i] + (1.f - a.alpha) * now : now;
                    if (cls == a.task) mem = cur[e];
                    a.qmem[i] = mem;
                    P[c + e] = mem;
                }
            }
        }
        __syncthreads();                                // this workgroup's row of P is complete (same threads re-read their own stores
    }                                                   // below only through the strided loop: make every store visible first)
    float s = 0.f;
    for (int c = threadIdx.x; c < d; c += 256) { float t = tanhf(P[c]); s += t * t; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    const float inv = 1.0f / fmaxf(sqrtf((sh[0] + sh[1]) + (sh[2] + sh[3])), 1e-12f);
    for (int c = threadIdx.x; c < d; c += 256) An[c] = tanhf(P[c]) * inv;
}

// retrieval of both heads in one launch: blockIdx.y = 0 question prototypes -> memory row S, 1 category prototypes -> row S + 1
struct Retrieve2 { const float *protos[2], *An[2], *pool[2]; long long* idx[2]; float* out_f32[2]; bf16_t* out_bf16[2]; int C[2]; };
__global__ __launch_bounds__(1024) void retrieve2_kernel(Retrieve2 r, long long sb, long long sb16, int B, int d) {
    // 16 waves per (sample, head): wave w scores classes w, w+16, ... -- five at a time, so the whole similarity row of the 80
    // categories is ONE round of independent loads per wave (with four waves it was five dependent rounds: 17 us of latency).
    // The arithmetic per class (and the norm of the sample, computed by the first four waves) is that of retrieve_kernel.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int h = blockIdx.y, C = r.C[h];
    float* tx = lds;
    float* sim = lds + d;
    __shared__ float part[4];
    __shared__ int best_sh;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* x = r.pool[h] + (size_t)b * d;
    const float* An = r.An[h];
    if (threadIdx.x < 256) {
        float s = 0.f;
        for (int c = threadIdx.x; c < d; c += 256) { float t = tanhf(x[c]); tx[c] = t; s += t * t; }
        s = wave_sum(s);
        if (lane == 0) part[wave] = s;
    }
    __syncthreads();
    const float nb = fmaxf(sqrtf((part[0] + part[1]) + (part[2] + part[3])), 1e-12f);
    constexpr int U = 5;
    for (int cls0 = wave; cls0 < C; cls0 += 16 * U) {
        float dot[U];
#pragma unroll
        for (int u = 0; u < U; ++u) dot[u] = 0.f;
        for (int c = lane * 4; c < d; c += 256) {
            const float4 t = *reinterpret_cast<const float4*>(tx + c);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cls = cls0 + 16 * u;
                if (cls < C) {
                    const float4 q = *reinterpret_cast<const float4*>(An + (size_t)cls * d + c);
                    dot[u] += (q.x * t.x + q.y * t.y) + (q.z * t.z + q.w * t.w);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cls = cls0 + 16 * u;
            const float v = wave_sum(dot[u]);
            if (lane == 0 && cls < C) sim[cls] = v / nb;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0;
        float bv = sim[0];
        for (int cls = 1; cls < C; ++cls)
            if (sim[cls] > bv) { bv = sim[cls]; best = cls; }
        best_sh = best;
        r.idx[h][b] = best;
    }
    __syncthreads();
    const float* sel = r.protos[h] + (size_t)best_sh * d;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        const float v = sel[c];
        if (r.out_f32[h]) r.out_f32[h][b * sb + c] = v;
        if (r.out_bf16[h]) r.out_bf16[h][b * sb16 + c] = f32_to_bf16(v);
    }
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
// Data parallel: the class statistics of both heads as ONE buffer for a single all-reduce, [sums Q | counts Q | sums V | counts V]
// with sums = mean * max(count, 1) -- and the way back (means over the global batch).  One launch each instead of a dozen
// element-wise torch launches in the middle of the forward.
__global__ void proto_stats_kernel(float* __restrict__ curQ, float* __restrict__ numQ, float* __restrict__ curV, float* __restrict__ numV,
                                   float* __restrict__ packed, int CQ, int CV, int d, int unpack) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = CQ * d, nv = CV * d;
    float* pq = packed; float* pnq = packed + nq; float* pv = pnq + CQ; float* pnv = pv + nv;
    if (i < nq) {
        const float c = fmaxf(unpack ? pnq[i / d] : numQ[i / d], 1.f);
        if (unpack) curQ[i] = pq[i] / c; else pq[i] = curQ[i] * c;
    } else if (i < nq + nv) {
        const int j = i - nq;
        const float c = fmaxf(unpack ? pnv[j / d] : numV[j / d], 1.f);
        if (unpack) curV[j] = pv[j] / c; else pv[j] = curV[j] * c;
    } else if (i < nq + nv + CQ) {
        const int j = i - nq - nv;
        if (unpack) numQ[j] = pnq[j]; else pnq[j] = numQ[j];
    } else if (i < nq + nv + CQ + CV) {
        const int j = i - nq - nv - CQ;
        if (unpack) numV[j] = pnv[j]; else pnv[j] = numV[j];
    }
}
extern "C" int vlt5_proto_stats_pack(float* curQ, float* numQ, float* curV, float* numV, float* packed, int CQ, int CV, int d,
                                     int unpack, void* stream) {
    if (!curQ || !numQ || !curV || !numV || !packed || CQ <= 0 || CV <= 0 || d <= 0) return VLT5_ERR_ARG;
    const int n = (CQ + CV) * (d + 1);
    hipLaunchKernelGGL(proto_stats_kernel, dim3((n + 255) / 256), dim3(256), 0, ST, curQ, numQ, curV, numV, packed, CQ, CV, d, unpack ? 1 : 0);
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

// The whole SS/SI prototype head of one forward (SURVEY 8(b) `vlt5_proto_head_fwd`; VL-T5/src/modeling_t5_our.py:583-615 with
// calculate_current_prototype :500-511, update_prototype :465-498, cosine_similarity_multi :434-462) in three launches: token
// pooling, one workgroup per prototype row (class mean of the batch + state update + normalised copy), retrieval of both heads.
// The per-task control flow (first batch of a task, which memory tensor) stays with the caller, as plain arguments.
extern "C" int vlt5_proto_head_fwd(const vlt5_proto_head_desc* h, void* stream) {
    if (!h || !h->hidden || !h->poolQ || !h->poolV || !h->Qproto || !h->Vproto || !h->scratch || !h->idxQ || !h->idxV) return VLT5_ERR_ARG;
    if (h->B <= 0 || h->S <= 0 || h->CQ <= 0 || h->CV <= 0 || h->split <= 0) return VLT5_ERR_ARG;
    if ((h->d & 3) || h->d > 2048 || (h->hidden_sb & 3)) return VLT5_ERR_ALIGN;
    if (h->phase < 0 || h->phase > 2 || (h->phase != 0 && (!h->packed || !h->update))) return VLT5_ERR_ARG;
    if (h->update) {
        if ((h->phase != 2 && (!h->onehotQ || !h->onehotV)) || !h->Qnum || !h->Vnum || h->task < 0 || h->task >= h->CQ) return VLT5_ERR_ARG;
        if (!h->first && h->task != 0 && !h->qmem) return VLT5_ERR_ARG;
    }
    int rc = VLT5_OK;
    if (h->phase != 2) rc = vlt5_proto_pool(h->hidden, h->hidden_sb, h->B, h->S, h->d, h->split, h->poolQ, h->poolV, stream);
    if (rc) return rc;
    HeadArgs a;
    a.packed = h->packed; a.phase = h->phase;
    a.poolQ = h->poolQ; a.poolV = h->poolV; a.onehotQ = h->onehotQ; a.onehotV = h->onehotV;
    a.Qp = h->Qproto; a.Vp = h->Vproto; a.Qnum = h->Qnum; a.Vnum = h->Vnum; a.qmem = h->qmem;
    a.AnQ = h->scratch; a.AnV = h->scratch + (size_t)h->CQ * h->d;
    a.qmem_init = h->qmem_initialised; a.first = h->first; a.task = h->task; a.update = h->update; a.alpha = h->alpha; a.beta = h->beta;
    a.B = h->B; a.CQ = h->CQ; a.CV = h->CV; a.d = h->d;
    hipLaunchKernelGGL(proto_row_kernel, dim3(h->CQ + h->CV), dim3(256), h->B * sizeof(float), ST, a);
    LAUNCH_CHECK();
    if (h->phase == 1) return VLT5_OK;                  // (the caller all-reduces `packed`, then calls again with phase = 2)
    Retrieve2 r;
    r.protos[0] = h->Qproto; r.protos[1] = h->Vproto; r.An[0] = a.AnQ; r.An[1] = a.AnV; r.pool[0] = h->poolQ; r.pool[1] = h->poolV;
    r.idx[0] = h->idxQ; r.idx[1] = h->idxV; r.C[0] = h->CQ; r.C[1] = h->CV;
    r.out_f32[0] = h->out_f32; r.out_f32[1] = h->out_f32 ? h->out_f32 + h->d : nullptr;
    r.out_bf16[0] = (bf16_t*)h->out_bf16; r.out_bf16[1] = h->out_bf16 ? (bf16_t*)h->out_bf16 + h->d : nullptr;
    const int cmax = h->CQ > h->CV ? h->CQ : h->CV;
    hipLaunchKernelGGL(retrieve2_kernel, dim3(h->B, 2), dim3(1024), (h->d + cmax) * sizeof(float), ST, r, h->out_sb, h->out_sb_bf16, h->B, h->d);
    LAUNCH_CHECK();
    return VLT5_OK;
}
