// LDS fragment-read throughput probe (gfx950): cycles per wave-instruction of the reads the GEMM main loops issue, with their
// real address patterns -- ds_read_b128 on a swizzled row-major tile, ds_read_b64_tr_b16 on a k-major tile (R = 64 / 128 / 256),
// plain ds_read_b64 at the same addresses -- for 1, 4 and 8 waves per workgroup (one workgroup per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o build/lds_probe && build/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

__device__ __forceinline__ s16x4_t tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
}
// plain reads as volatile asm (the compiler would hoist loop-invariant LDS loads); the caller waits with lgkmcnt(0) before consuming
__device__ __forceinline__ s16x4_t b64_read(const char* p) {
    s16x4_t v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p));
    return v;
}
__device__ __forceinline__ s16x8_t b128_read(const char* p) {
    s16x8_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p));
    return v;
}

template <int R>
__device__ __forceinline__ int km_swz(int k) {
    return R >= 128 ? (((k & 3) << 1) | (((k >> 3) & 1) << 3)) : ((((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2));
}
// address of the low transpose read of fragment (r0, ks) in a k-major [64][R] tile (gemm_kernel.h frag_km)
template <int R>
__device__ __forceinline__ int km_addr(int r0, int ks, int lane, int swz_on) {
    const int g = lane >> 4, i = lane & 15;
    const int k = ks * 32 + g * 8 + (i >> 2);
    const int cb = r0 * 2 + (i & 3) * 8;
    return k * (R * 2) + (((cb >> 4) ^ (swz_on ? km_swz<R>(k) : 0)) << 4) + (cb & 15);
}
__device__ __forceinline__ int rm_addr(int row, int kchunk, int swz_on) { return row * 128 + ((kchunk ^ (swz_on ? (row & 7) : 0)) << 4); }

// MODE 0: b128 row-major; 1: tr_b64 km R; 2: plain b64 at the km addresses; NF fragments per iteration (independent reads in flight)
template <int MODE, int R, int NF>
__global__ void probe(unsigned long long* out, int iters, int swz_on) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<int*>(smem)[i] = i;
    __syncthreads();
    int addr[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        if (MODE == 0) addr[f] = rm_addr(((wave & 1) * (R / 2) + f * 16) % R + (lane & 15), (lane >> 4), swz_on);
        else           addr[f] = km_addr<R>(((wave & 1) * (R / 2) + f * 16) % R, 0, lane, swz_on);
    }
    int acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            s16x8_t v[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) v[f] = b128_read(smem + addr[f]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int f = 0; f < NF; ++f) acc ^= v[f][0] ^ v[f][7];
        } else if (MODE == 1) {
            s16x4_t lo[NF], hi[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) { lo[f] = tr_read(smem + addr[f]); hi[f] = tr_read(smem + addr[f] + 4 * (R * 2)); }
#pragma unroll
            for (int f = 0; f < NF; ++f) acc ^= lo[f][0] ^ hi[f][3];
        } else {
            s16x4_t lo[NF], hi[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) { lo[f] = b64_read(smem + addr[f]); hi[f] = b64_read(smem + addr[f] + 4 * (R * 2)); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int f = 0; f < NF; ++f) acc ^= lo[f][0] ^ hi[f][3];
        }
        // ks = 1 half: +32 k rows (k-major) / +4 chunks (row-major): flip between the halves so the addresses are not loop invariant
#pragma unroll
        for (int f = 0; f < NF; ++f) addr[f] ^= (MODE == 0) ? 64 : 32 * (R * 2);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    if (acc == 0x12345678) out[1023] = acc;
}

template <int MODE, int R, int NF>
void run(const char* name, int waves, int swz_on) {
    unsigned long long* d;
    hipMalloc(&d, 1024 * 8);
    hipMemset(d, 0, 1024 * 8);
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, R, NF>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((probe<MODE, R, NF>), dim3(1), dim3(waves * 64), 65536, 0, d, iters, swz_on);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(16);
    hipMemcpy(h.data(), d, 16 * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    const int per_iter = NF * (MODE == 0 ? 1 : 2);
    const double clk_per_instr_wave = (double)mx / iters / per_iter;                  // as one wave sees it
    const double clk_per_instr_cu = (double)mx / iters / (per_iter * waves);          // LDS time per wave-instruction
    const double bytes = (MODE == 0 ? 1024.0 : 512.0);
    printf("%-28s R=%3d NF=%d swz=%d waves=%d : %7.2f clk/instr/wave  %6.2f clk/instr (CU)  %6.1f B/clk\n", name, R, NF, swz_on, waves,
           clk_per_instr_wave, clk_per_instr_cu, bytes / clk_per_instr_cu);
    hipFree(d);
}

int main() {
    for (int waves : {1, 4, 8}) {
        for (int swz : {1, 0}) {
            run<0, 128, 8>("ds_read_b128 row-major", waves, swz);
            run<1, 64, 4>("ds_read_b64_tr_b16 k-major", waves, swz);
            run<1, 128, 8>("ds_read_b64_tr_b16 k-major", waves, swz);
            run<1, 256, 8>("ds_read_b64_tr_b16 k-major", waves, swz);
            run<2, 128, 8>("ds_read_b64 (plain) k-major", waves, swz);
            run<2, 256, 8>("ds_read_b64 (plain) k-major", waves, swz);
        }
    }
    return 0;
}
