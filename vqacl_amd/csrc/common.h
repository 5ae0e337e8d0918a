// Shared device helpers for the VL-T5 gfx950 kernels (wave64, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;                                             // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;         // one MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;          // one MFMA C/D fragment

#define VLT5_OK 0
#define VLT5_ERR_ARG 1001          // bad argument (null pointer, unsupported size)
#define VLT5_ERR_ALIGN 1002        // a contiguous dimension is not a multiple of 8 elements

#define WAVE 64
#ifndef LNB_MAXBLK
#define LNB_MAXBLK 640             // workgroups of a norm backward = weight-gradient partials per norm (engine scratch: 64 x LNB_MAXBLK x d).
                                   // 320 until round 5: ten waves per CU instead of five keep more of the cold rows in flight (same-box A/B
                                   // 8.56 -> 8.53 ms per step twice; 1120: slower, the partials' reduction grows)
#endif

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// f32 -> bf16, round to nearest even: the native conversion (clang lowers `(__bf16)x` to v_cvt_pk_bf16_f32 on gfx950, one
// instruction per PAIR; the integer emulation it replaces cost ~8 VALU instructions per element and dominated every bf16 epilogue)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16_native2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16_native2_t v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

// ---- counter-based dropout mask ---------------------------------------------------------------
// keep(idx) is a pure function of (seed, idx): forward and backward regenerate the same mask, no
// mask tensor is stored.  One 32-bit hash serves two neighbouring elements (16 bits each).
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
__host__ __device__ __forceinline__ uint32_t drop_thr16(float p) {    // P(drop) = thr/65536
    float t = p * 65536.0f + 0.5f;
    return t <= 0.f ? 0u : (t >= 65535.f ? 65535u : (uint32_t)t);
}
__host__ __device__ __forceinline__ float drop_scale(uint32_t thr16) { return 65536.0f / (float)(65536u - thr16); }
__host__ __device__ __forceinline__ bool drop_keep(uint32_t seed, uint32_t idx, uint32_t thr16) {
    uint32_t h = mix32(((idx >> 1) * 0x9E3779B1u) ^ seed);
    uint32_t v = (idx & 1u) ? (h >> 16) : (h & 0xffffu);
    return v >= thr16;
}
// keep flags of FOUR consecutive elements idx .. idx+3, bit-identical to drop_keep() per element: with an even idx (every call site
// whose row length is even) the four share two hashes -- half the integer work of the dropout epilogues
__host__ __device__ __forceinline__ void drop_keep4(uint32_t seed, uint32_t idx, uint32_t thr16, bool (&k)[4]) {
    if ((idx & 1u) == 0u) {
        const uint32_t h0 = mix32(((idx >> 1) * 0x9E3779B1u) ^ seed), h1 = mix32((((idx >> 1) + 1u) * 0x9E3779B1u) ^ seed);
        k[0] = (h0 & 0xffffu) >= thr16; k[1] = (h0 >> 16) >= thr16; k[2] = (h1 & 0xffffu) >= thr16; k[3] = (h1 >> 16) >= thr16;
    } else {
        for (int e = 0; e < 4; ++e) k[e] = drop_keep(seed, idx + e, thr16);
    }
}
__host__ __device__ __forceinline__ uint32_t site_seed(uint32_t base, uint32_t site) {
    return mix32(base ^ (site * 0x632BE5ABu + 0x9E3779B9u));
}

// exp for the softmax paths: v_exp_f32 on x * log2(e) (~1 ulp of the result for the score ranges of a softmax, far below the bf16
// rounding of the probabilities that follows); expf() expands to ~20 instructions per element and made the attention core VALU-bound
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }

// ---- output stores that another KERNEL reads ------------------------------------------------------
// Written through the L2 (sc0 sc1) instead of left dirty in it: the 8 XCD-private L2s are not coherent with each other, so whatever
// a kernel leaves dirty is written back when it ENDS, after its last workgroup -- launch-to-launch time minus workgroup span grew
// with the output size (~8 us for a 27 MB GEMM output, tools/gemm_timeline.py).  Write-through spreads that traffic over the
// kernel's life.  (Non-temporal stores measured slower; -DVLT5_WT_STORE=0 restores plain stores for A/B runs.)
#ifndef VLT5_WT_STORE
#define VLT5_WT_STORE 1
#endif
typedef __attribute__((ext_vector_type(4))) unsigned vlt5_u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned vlt5_u32x2_t;
__device__ __forceinline__ void store_wt16(void* dst, uint4 v) {
#if VLT5_WT_STORE
    const vlt5_u32x4_t q = {v.x, v.y, v.z, v.w};
    // (s_nop 1: a store of more than 8 bytes reads its data late -- a VALU write to those registers needs two wait states after it;
    // the compiler pads its own stores, it cannot see into this one)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(q) : "memory");
#else
    *reinterpret_cast<uint4*>(dst) = v;
#endif
}
__device__ __forceinline__ void store_wt8(void* dst, uint2 v) {
#if VLT5_WT_STORE
    const vlt5_u32x2_t q = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(q) : "memory");
#else
    *reinterpret_cast<uint2*>(dst) = v;
#endif
}
__device__ __forceinline__ void store_wt16f(float* dst, float4 v) {
    store_wt16(dst, make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)));
}

// ---- wave reductions ----------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Kernels with more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize, per kernel AND per device.  `done` is
// the caller's per-kernel mask of device ordinals that already have it (one relaxed load on the hot path); safe from several host
// threads and after hipSetDevice: nothing about the launch path is tied to the first caller.
#include <atomic>
static inline int vlt5_lds_optin(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return VLT5_ERR_ARG;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return VLT5_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done.fetch_or(bit, std::memory_order_release);
    return VLT5_OK;
}

#define HIP_RET(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return (int)e__; } while (0)
#define LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
