// Optimizer-side kernels: global gradient norm, fused clip + AdamW over the flat parameter buffer (also refreshes the bf16
// shadow the GEMMs read), fp32 -> bf16 cast.  HBM-bound: AdamW moves 28 B/param (+2 B for the shadow).
#include "common.h"
#include "vlt5_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4v;     // native vector type accepted by the non-temporal builtins
__device__ __forceinline__ float4 nt_load4(const float* p) {
    f32x4v v = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void nt_store4(float* p, float a, float b, float c, float d) {
    f32x4v v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4v*>(p));
}

// four gradients starting at element i: f32 as they are, or (G16) bf16 times `gs` -- exactly what vlt5_cast_f32 would have written
template <bool G16>
__device__ __forceinline__ float4 grad4(const void* __restrict__ g, long long i, float gs, bool nt) {
    if (G16) {
        const uint2 a = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(g) + i);
        return make_float4(__uint_as_float(a.x << 16) * gs, __uint_as_float(a.x & 0xffff0000u) * gs,
                           __uint_as_float(a.y << 16) * gs, __uint_as_float(a.y & 0xffff0000u) * gs);
    }
    const float* f = reinterpret_cast<const float*>(g) + i;
    return nt ? nt_load4(f) : *reinterpret_cast<const float4*>(f);
}
template <bool G16>
__device__ __forceinline__ float grad1(const void* __restrict__ g, long long i, float gs) {
    return G16 ? bf16_to_f32(reinterpret_cast<const bf16_t*>(g)[i]) * gs : reinterpret_cast<const float*>(g)[i];
}

template <bool G16>
__global__ __launch_bounds__(256) void sqnorm_kernel(const void* __restrict__ g, float gs, long long n, float* __restrict__ partial) {
    __shared__ float sh[4];
    float s = 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            float4 v = grad4<G16>(g, i, gs, false);
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        } else {
            for (long long k = i; k < n; ++k) { const float x = grad1<G16>(g, k, gs); s += x * x; }
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void sqnorm_final_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ total, int accum) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 64) s += partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) total[0] = accum ? total[0] + s : s;
}

// total = sum(a[0..na)) + sum(b[0..nb)): one workgroup of 1024 lanes, 16-byte loads, four independent partial sums per lane (the
// sums are latency-bound: 55 k slots read one dependent word at a time by 256 lanes took 87 us), then lanes and waves in a fixed order
__global__ __launch_bounds__(1024) void gnorm_final_kernel(const float* __restrict__ a, long long na, const float* __restrict__ b, long long nb,
                                                           float* __restrict__ total) {
    __shared__ float sh[16];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const long long na4 = na >> 2;                        // (a is 16-byte aligned: a torch allocation)
    const float4* a4 = reinterpret_cast<const float4*>(a);
    long long i = threadIdx.x;
    for (; i + 3 * 1024 < na4; i += 4 * 1024) {
        const float4 u0 = a4[i], u1 = a4[i + 1024], u2 = a4[i + 2048], u3 = a4[i + 3072];
        s0 += (u0.x + u0.y) + (u0.z + u0.w); s1 += (u1.x + u1.y) + (u1.z + u1.w);
        s2 += (u2.x + u2.y) + (u2.z + u2.w); s3 += (u3.x + u3.y) + (u3.z + u3.w);
    }
    for (; i < na4; i += 1024) { const float4 u = a4[i]; s0 += (u.x + u.y) + (u.z + u.w); }
    for (long long j = (na4 << 2) + threadIdx.x; j < na; j += 1024) s1 += a[j];
    for (long long j = threadIdx.x; j < nb; j += 1024) s2 += b[j];
    float s = wave_sum((s0 + s1) + (s2 + s3));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += sh[w];
        total[0] = t;
    }
}

template <bool G16>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const void* __restrict__ g, float gs, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ pb, long long n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                    const float* __restrict__ total_sq, float max_norm, int hf_mode) {
    float clip = 1.f;
    if (total_sq) clip = fminf(1.f, max_norm / (sqrtf(total_sq[0]) + 1e-6f));
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        const int cnt = (i + 3 < n) ? 4 : (int)(n - i);
        float pv[4], gv[4], mv[4], vv[4];
        if (cnt == 4) {
            // streamed once per step: non-temporal so 3.6 GB of optimizer state do not evict the weights' bf16 shadow from L2/MALL
            float4 a = nt_load4(p + i), b = grad4<G16>(g, i, gs, true), c = nt_load4(m + i), e = nt_load4(v + i);
            pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w; gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
            mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w; vv[0] = e.x; vv[1] = e.y; vv[2] = e.z; vv[3] = e.w;
        } else {
            for (int k = 0; k < cnt; ++k) { pv[k] = p[i + k]; gv[k] = grad1<G16>(g, i + k, gs); mv[k] = m[i + k]; vv[k] = v[i + k]; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < cnt) {
                const float gr = gv[k] * clip;
                mv[k] = b1 * mv[k] + (1.f - b1) * gr;
                vv[k] = b2 * vv[k] + (1.f - b2) * gr * gr;
                if (hf_mode) {           // transformers.AdamW: denom = sqrt(v)+eps ; step = lr*sqrt(bc2)/bc1 ; then decay
                    const float step = lr * sqrtf(bc2) / bc1;
                    pv[k] = pv[k] - step * (mv[k] / (sqrtf(vv[k]) + eps));
                    if (wd > 0.f) pv[k] = pv[k] - lr * wd * pv[k];
                } else {                 // torch.optim.AdamW: decay first ; denom = sqrt(v)/sqrt(bc2)+eps ; step = lr/bc1
                    if (wd > 0.f) pv[k] = pv[k] * (1.f - lr * wd);
                    pv[k] = pv[k] - (lr / bc1) * (mv[k] / (sqrtf(vv[k]) / sqrtf(bc2) + eps));
                }
            }
        }
        if (cnt == 4) {
            nt_store4(p + i, pv[0], pv[1], pv[2], pv[3]);
            nt_store4(m + i, mv[0], mv[1], mv[2], mv[3]);
            nt_store4(v + i, vv[0], vv[1], vv[2], vv[3]);
            if (pb) {
                uint2 pk;
                pk.x = pack_bf16x2(pv[0], pv[1]);
                pk.y = pack_bf16x2(pv[2], pv[3]);
                *reinterpret_cast<uint2*>(pb + i) = pk;
            }
        } else {
            for (int k = 0; k < cnt; ++k) {
                p[i + k] = pv[k]; m[i + k] = mv[k]; v[i + k] = vv[k];
                if (pb) pb[i + k] = f32_to_bf16(pv[k]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x * 8;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 7 < n) {
            float4 a = *reinterpret_cast<const float4*>(src + i), b = *reinterpret_cast<const float4*>(src + i + 4);
            uint4 o;
            o.x = pack_bf16x2(a.x, a.y); o.y = pack_bf16x2(a.z, a.w); o.z = pack_bf16x2(b.x, b.y); o.w = pack_bf16x2(b.z, b.w);
            *reinterpret_cast<uint4*>(dst + i) = o;
        } else {
            for (long long k = i; k < n; ++k) dst[k] = f32_to_bf16(src[k]);
        }
    }
}

// dst = scale * float(src): the way back from a bf16 gradient all-reduce (scale = 1/world averages)
__global__ __launch_bounds__(256) void uncast_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long long n, float scale) {
    const long long stride = (long long)gridDim.x * blockDim.x * 8;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 7 < n) {
            const uint4 a = *reinterpret_cast<const uint4*>(src + i);
            const uint32_t w[4] = {a.x, a.y, a.z, a.w};
            float o[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[2 * k] = __uint_as_float(w[k] << 16) * scale; o[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u) * scale; }
            *reinterpret_cast<float4*>(dst + i) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(dst + i + 4) = make_float4(o[4], o[5], o[6], o[7]);
        } else {
            for (long long k = i; k < n; ++k) dst[k] = bf16_to_f32(src[k]) * scale;
        }
    }
}

__global__ __launch_bounds__(256) void scale_add_kernel(float* __restrict__ dst, const float* __restrict__ src, float a, float b,
                                                        long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = a * dst[i] + b * src[i];
}

__global__ __launch_bounds__(256) void drop_cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n,
                                                        uint32_t thr, uint32_t seed) {
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    const float dsc = drop_scale(thr);
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        float4 a = *reinterpret_cast<const float4*>(src + i);
        float o[4] = {a.x, a.y, a.z, a.w};
        if (thr) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = drop_keep(seed, (uint32_t)(i + k), thr) ? o[k] * dsc : 0.f;
        }
        uint2 pk;
        pk.x = pack_bf16x2(o[0], o[1]);
        pk.y = pack_bf16x2(o[2], o[3]);
        *reinterpret_cast<uint2*>(dst + i) = pk;
    }
}

inline int grid_for(long long n, int per_thread) {
    long long b = (n / per_thread + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;         // 256 CUs x 8 blocks, grid-stride beyond that
    return (int)b;
}

}  // namespace

#define ST ((hipStream_t)stream)
extern "C" int vlt5_sqnorm_blocks(long long n) { return grid_for(n, 4); }
extern "C" int vlt5_sqnorm(const float* g, long long n, float* partial, float* total_sq, int accum_total, void* stream) {
    if (!g || !partial || !total_sq || n <= 0) return VLT5_ERR_ARG;
    if (((uintptr_t)g) & 15) return VLT5_ERR_ALIGN;
    int nblk = grid_for(n, 4);
    hipLaunchKernelGGL(sqnorm_kernel<false>, dim3(nblk), dim3(256), 0, ST, (const void*)g, 1.f, n, partial);
    LAUNCH_CHECK();
    if (accum_total == 2) return VLT5_OK;           // partials only: the caller sums several ranges with one vlt5_gnorm_finish
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(64), 0, ST, partial, nblk, total_sq, accum_total);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_gnorm_finish(const float* partials, long long nslots, const float* grads, const long long* range_off,
                                 const long long* range_n, int nranges, float* scratch, float* total_sq, void* stream) {
    if (!partials || nslots <= 0 || !scratch || !total_sq || nranges < 0 || nranges > 4 || (nranges > 0 && !grads)) return VLT5_ERR_ARG;
    if (((uintptr_t)partials) & 15) return VLT5_ERR_ALIGN;
    long long used = 0;
    for (int i = 0; i < nranges; ++i) {
        if (range_n[i] <= 0) continue;
        if (range_off[i] & 3) return VLT5_ERR_ALIGN;
        const int nblk = grid_for(range_n[i], 4);
        hipLaunchKernelGGL(sqnorm_kernel<false>, dim3(nblk), dim3(256), 0, ST, (const void*)(grads + range_off[i]), 1.f, range_n[i],
                           scratch + used);
        LAUNCH_CHECK();
        used += nblk;
    }
    hipLaunchKernelGGL(gnorm_final_kernel, dim3(1), dim3(1024), 0, ST, partials, nslots, (const float*)scratch, used, total_sq);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_sqnorm_g16(const void* g_bf16, float g_scale, long long n, float* partial, float* total_sq, int accum_total,
                               void* stream) {
    if (!g_bf16 || !partial || !total_sq || n <= 0) return VLT5_ERR_ARG;
    if (((uintptr_t)g_bf16) & 7) return VLT5_ERR_ALIGN;
    int nblk = grid_for(n, 4);
    hipLaunchKernelGGL(sqnorm_kernel<true>, dim3(nblk), dim3(256), 0, ST, g_bf16, g_scale, n, partial);
    LAUNCH_CHECK();
    if (accum_total == 2) return VLT5_OK;           // partials only (see vlt5_sqnorm)
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(64), 0, ST, partial, nblk, total_sq, accum_total);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, long long n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, const float* total_sq, float max_norm,
                               int hf_mode, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return VLT5_ERR_ARG;
    if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return VLT5_ERR_ALIGN;
    if (p_bf16 && (((uintptr_t)p_bf16) & 7)) return VLT5_ERR_ALIGN;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel<false>, dim3(grid_for(n, 4)), dim3(256), 0, ST, p, (const void*)g, 1.f, m, v, (bf16_t*)p_bf16, n, lr,
                       beta1, beta2, eps, weight_decay, bc1, bc2, total_sq, max_norm, hf_mode);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_adamw_step_g16(float* p, const void* g_bf16, float g_scale, float* m, float* v, void* p_bf16, long long n, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, int step, const float* total_sq,
                                   float max_norm, int hf_mode, void* stream) {
    if (!p || !g_bf16 || !m || !v || n <= 0 || step < 1) return VLT5_ERR_ARG;
    if ((((uintptr_t)p) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return VLT5_ERR_ALIGN;
    if ((((uintptr_t)g_bf16) & 7) || (p_bf16 && (((uintptr_t)p_bf16) & 7))) return VLT5_ERR_ALIGN;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel<true>, dim3(grid_for(n, 4)), dim3(256), 0, ST, p, g_bf16, g_scale, m, v, (bf16_t*)p_bf16, n, lr,
                       beta1, beta2, eps, weight_decay, bc1, bc2, total_sq, max_norm, hf_mode);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_cast_bf16(const float* src, void* dst_bf16, long long n, void* stream) {
    if (!src || !dst_bf16 || n <= 0) return VLT5_ERR_ARG;
    if ((((uintptr_t)src) | ((uintptr_t)dst_bf16)) & 15) return VLT5_ERR_ALIGN;
    hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n, 8)), dim3(256), 0, ST, src, (bf16_t*)dst_bf16, n);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_cast_f32(const void* src_bf16, float* dst, long long n, float scale, void* stream) {
    if (!src_bf16 || !dst || n <= 0) return VLT5_ERR_ARG;
    if ((((uintptr_t)src_bf16) | ((uintptr_t)dst)) & 15) return VLT5_ERR_ALIGN;
    hipLaunchKernelGGL(uncast_kernel, dim3(grid_for(n, 8)), dim3(256), 0, ST, (const bf16_t*)src_bf16, dst, n, scale);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_scale_add(float* dst, const float* src, float a, float b, long long n, void* stream) {
    if (!dst || !src || n <= 0) return VLT5_ERR_ARG;
    hipLaunchKernelGGL(scale_add_kernel, dim3(grid_for(n, 1)), dim3(256), 0, ST, dst, src, a, b, n);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_drop_cast(const float* src, void* dst_bf16, long long rows, int cols, float drop_p, uint32_t drop_seed,
                              void* stream) {
    if (!src || !dst_bf16 || rows <= 0 || cols <= 0) return VLT5_ERR_ARG;
    if (cols & 3) return VLT5_ERR_ALIGN;
    const long long n = rows * cols;
    hipLaunchKernelGGL(drop_cast_kernel, dim3(grid_for(n, 4)), dim3(256), 0, ST, src, (bf16_t*)dst_bf16, n,
                       drop_p > 0.f ? drop_thr16(drop_p) : 0u, drop_seed);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_abi_version(void) { return VLT5_ABI_VERSION; }
// per translation unit (an experiment flag may be given to one file only): the units that hold wrong-result switches report their own
int vlt5_build_flags_engine();
int vlt5_build_flags_attn();
extern "C" int vlt5_build_flags(void) {
    int f = vlt5_build_flags_engine() | vlt5_build_flags_attn();
#if defined(GEMM_TIMELINE) || defined(ATTN_TIMELINE) || defined(DECLIN_TIMELINE)
    f |= VLT5_BUILD_TIMELINE;
#endif
#if defined(ENC_DGRAD_HOT_A)
    f |= VLT5_BUILD_ENC_DGRAD_HOT_A;
#endif
#if defined(ATTN_BWD_NO_STORE)
    f |= VLT5_BUILD_ATTN_BWD_NO_STORE;
#endif
    return f;
}
