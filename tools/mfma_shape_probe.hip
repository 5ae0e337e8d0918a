// MFMA shape probe for the 8-wave 256 x 256 GEMM k-step (review item: "v_mfma_f32_32x32x16_bf16 -- half the fragment reads per flop").
// A wave of that kernel owns a 128 x 64 output tile.  Per 64-deep k-step it needs (128 + 64) rows x 64 k of bf16 out of LDS whatever the
// MFMA shape: with v_mfma_f32_16x16x32_bf16 that is 8 + 4 row fragments x 2 k-halves = 24 ds_read_b128 and 64 MFMAs of 16 passes, with
// v_mfma_f32_32x32x16_bf16 4 + 2 row fragments x 4 k-quarters = 24 ds_read_b128 and 32 MFMAs of 32 passes -- the SAME bytes and the same
// matrix-pipe time; only the instruction count halves.  This probe runs exactly those two k-steps (8 waves per workgroup, one workgroup
// per CU, operands re-read from LDS every step, no global traffic) and reports shader cycles per k-step, i.e. what the main loop of the
// 256 x 256 kernel could gain from the other shape before any of its epilogue / lane-map rewrite.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o build/mfma_shape_probe && build/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int SHAPE>
__global__ __launch_bounds__(512) void kstep_probe(const uint4* __restrict__ seed, float* __restrict__ sink, long long* __restrict__ cycles, int steps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];                 // 2 stages x (256 + 256 rows) x 128 B = 128 KB, as in the GEMM
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 131072 / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = seed[i & 1023];
    __syncthreads();
    const int wm = w >> 2, wn = w & 3;                                         // 2 x 4 waves: 128 rows x 64 columns each
    const char* at = lds + wm * 128 * 128;                                     // this wave's A rows, [row][64 k] with 128-byte rows
    const char* bt = lds + 65536 + wn * 64 * 128;
    const long long t0 = (long long)__builtin_readcyclecounter();
    if constexpr (SHAPE == 16) {
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < steps; ++s) {
            const int stage = (s & 1) * 32768;                                   // alternate the halves of the operand regions like a 2-stage ring
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t fa[8], fb[4];
                const int slot = (ks * 4 + (lane >> 4)) ^ (lane & 7);                 // the GEMM's XOR-swizzled 16-byte slot of a row
#pragma unroll
                for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(at + (stage & 16383) + (i * 16 + (lane & 15)) * 128 + slot * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(bt + (stage & 8191) + (j * 16 + (lane & 15)) * 128 + slot * 16);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_s_barrier();                                        // one barrier per k-step, as in the kernel
        }
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 12345.678f) sink[threadIdx.x] = t;
    } else {
        f32x16_t acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int s = 0; s < steps; ++s) {
            const int stage = (s & 1) * 32768;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {                                     // four 16-deep quarters; lane l: row l & 31, 8 k at (l >> 5) * 8
                bf16x8_t fa[4], fb[2];
                const int slot = (kq * 2 + (lane >> 5)) ^ (((lane & 31) >> 1) & 7);      // conflict-free for 32 rows of 128 bytes per ds_read_b128 group
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(at + (stage & 16383) + (i * 32 + (lane & 31)) * 128 + slot * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(bt + (stage & 8191) + (j * 32 + (lane & 31)) * 128 + slot * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_s_barrier();
        }
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) t += acc[i][j][e];
        if (t == 12345.678f) sink[threadIdx.x] = t;
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    if (lane == 0) cycles[blockIdx.x * 8 + w] = t1 - t0;
}

template <int SHAPE>
static double run(const uint4* seed, float* sink, long long* cyc, int steps) {
    hipFuncSetAttribute((const void*)kstep_probe<SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kstep_probe<SHAPE>, dim3(256), dim3(512), 131072, 0, seed, sink, cyc, steps);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (long long v : h) s += (double)v;
    return s / h.size() / steps;
}

int main() {
    uint4* seed; float* sink; long long* cyc;
    hipMalloc(&seed, 1024 * 16); hipMalloc(&sink, 4096); hipMalloc(&cyc, 256 * 8 * 8);
    std::vector<unsigned> h(4096);
    for (auto& v : h) { const unsigned a = 0x3f80u + (rand() & 0x7f), b = 0xbf80u + (rand() & 0x7f); v = a | (b << 16); }     // bf16 values around +-1
    hipMemcpy(seed, h.data(), 1024 * 16, hipMemcpyHostToDevice);
    const int steps = 2000;
    for (int rep = 0; rep < 3; ++rep) {
        const double c16 = run<16>(seed, sink, cyc, steps), c32 = run<32>(seed, sink, cyc, steps);
        printf("k-step of a 128x64 wave tile, 8 waves / CU, 256 CUs: 16x16x32 %7.1f cycles (64 MFMAs + 24 ds_read_b128), 32x32x16 %7.1f cycles "
               "(32 MFMAs + 24 ds_read_b128): %+.1f %%   [matrix pipe alone: 2 waves x 64 x 16 = 2048 cycles]\n",
               c16, c32, 100.0 * (c32 - c16) / c16);
    }
    return 0;
}
