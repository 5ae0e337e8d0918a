// Gated-GELU FFN activation (t5-v1.1 style configs): HF T5DenseGatedActDense, h = dropout(gelu_new(x W0^T) * (x W1^T)).
// The two input projections run as ONE GEMM over the adjacent [wi_0; wi_1] operand (u = x [W0;W1]^T, [rows, 2*ff] bf16, kept for
// the backward); these two HBM-bound kernels turn u into the hidden activation and the hidden gradient back into du.
// t5-base / t5-large (every BASELINE configuration) use the ReLU FFN, whose activation lives in the GEMM epilogues instead.
#include "common.h"
#include "vlt5_hip.h"

namespace {

constexpr float kC = 0.7978845608028654f;      // sqrt(2/pi)
constexpr float kA = 0.044715f;

__device__ __forceinline__ float gelu_new(float x) { return 0.5f * x * (1.f + tanhf(kC * (x + kA * x * x * x))); }
__device__ __forceinline__ float gelu_new_grad(float x) {
    const float t = tanhf(kC * (x + kA * x * x * x));
    return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * kC * (1.f + 3.f * kA * x * x);
}

// one thread: 8 consecutive hidden columns of one row
__global__ __launch_bounds__(256) void glu_fwd_kernel(const bf16_t* __restrict__ u, bf16_t* __restrict__ h, long long rows, int ff,
                                                      uint32_t thr, uint32_t seed) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cpr = ff / 8;
    if (t >= rows * cpr) return;
    const long long r = t / cpr;
    const int c = (int)(t % cpr) * 8;
    const uint4 a = *reinterpret_cast<const uint4*>(u + r * 2 * ff + c);
    const uint4 b = *reinterpret_cast<const uint4*>(u + r * 2 * ff + ff + c);
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    const float ds = drop_scale(thr);
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float x0 = bf16_to_f32((bf16_t)((aw[k] >> (16 * e)) & 0xffffu));
            const float x1 = bf16_to_f32((bf16_t)((bw[k] >> (16 * e)) & 0xffffu));
            float y = gelu_new(x0) * x1;
            if (thr) y = drop_keep(seed, (uint32_t)(r * ff + c + 2 * k + e), thr) ? y * ds : 0.f;
            v[e] = y;
        }
        o[k] = pack_bf16x2(v[0], v[1]);
    }
    *reinterpret_cast<uint4*>(h + r * ff + c) = make_uint4(o[0], o[1], o[2], o[3]);
}

// du[:, :ff] = dh * keep * u1 * gelu_new'(u0),  du[:, ff:] = dh * keep * gelu_new(u0)
__global__ __launch_bounds__(256) void glu_bwd_kernel(const bf16_t* __restrict__ dh, const bf16_t* __restrict__ u, bf16_t* __restrict__ du,
                                                      long long rows, int ff, uint32_t thr, uint32_t seed) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cpr = ff / 8;
    if (t >= rows * cpr) return;
    const long long r = t / cpr;
    const int c = (int)(t % cpr) * 8;
    const uint4 a = *reinterpret_cast<const uint4*>(u + r * 2 * ff + c);
    const uint4 b = *reinterpret_cast<const uint4*>(u + r * 2 * ff + ff + c);
    const uint4 g = *reinterpret_cast<const uint4*>(dh + r * ff + c);
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w}, gw[4] = {g.x, g.y, g.z, g.w};
    const float ds = drop_scale(thr);
    uint32_t o0[4], o1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v0[2], v1[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float x0 = bf16_to_f32((bf16_t)((aw[k] >> (16 * e)) & 0xffffu));
            const float x1 = bf16_to_f32((bf16_t)((bw[k] >> (16 * e)) & 0xffffu));
            float d = bf16_to_f32((bf16_t)((gw[k] >> (16 * e)) & 0xffffu));
            if (thr) d = drop_keep(seed, (uint32_t)(r * ff + c + 2 * k + e), thr) ? d * ds : 0.f;
            v0[e] = d * x1 * gelu_new_grad(x0);
            v1[e] = d * gelu_new(x0);
        }
        o0[k] = pack_bf16x2(v0[0], v0[1]);
        o1[k] = pack_bf16x2(v1[0], v1[1]);
    }
    *reinterpret_cast<uint4*>(du + r * 2 * ff + c) = make_uint4(o0[0], o0[1], o0[2], o0[3]);
    *reinterpret_cast<uint4*>(du + r * 2 * ff + ff + c) = make_uint4(o1[0], o1[1], o1[2], o1[3]);
}

}  // namespace

extern "C" int vlt5_glu_fwd(const void* u_bf16, void* h_bf16, long long rows, int ff, float drop_p, uint32_t drop_seed, void* stream) {
    if (!u_bf16 || !h_bf16 || rows <= 0 || ff <= 0) return VLT5_ERR_ARG;
    if (ff & 7) return VLT5_ERR_ALIGN;
    const long long n = rows * (ff / 8);
    hipLaunchKernelGGL(glu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)u_bf16,
                       (bf16_t*)h_bf16, rows, ff, drop_p > 0.f ? drop_thr16(drop_p) : 0u, drop_seed);
    LAUNCH_CHECK();
    return VLT5_OK;
}

extern "C" int vlt5_glu_bwd(const void* dh_bf16, const void* u_bf16, void* du_bf16, long long rows, int ff, float drop_p,
                            uint32_t drop_seed, void* stream) {
    if (!dh_bf16 || !u_bf16 || !du_bf16 || rows <= 0 || ff <= 0) return VLT5_ERR_ARG;
    if (ff & 7) return VLT5_ERR_ALIGN;
    const long long n = rows * (ff / 8);
    hipLaunchKernelGGL(glu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dh_bf16,
                       (const bf16_t*)u_bf16, (bf16_t*)du_bf16, rows, ff, drop_p > 0.f ? drop_thr16(drop_p) : 0u, drop_seed);
    LAUNCH_CHECK();
    return VLT5_OK;
}
