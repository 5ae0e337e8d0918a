// EXPERIMENT (not part of libvlt5_hip.so): the first half of the planned (sample, head-group) fused attention sublayer --
// T5 RMS-norm of one sample's rows into an LDS-resident bf16 A-panel, then the q|k|v projection of ONE head group (a
// [NCH*192, 768] slice of the fused QKV weight) out of that panel with the weight streamed by LDS-DMA.  One workgroup of 8 waves
// per (sample, head group): grid = B * 3 for t5-base.  Built and timed by tools/ln_qkv_probe.py against vlt5_layernorm_fwd +
// vlt5_gemm_bf16 (the two launches it would replace).  See DESIGN.md, "Attention design".
#include "../common.h"

namespace {

constexpr int D = 768;                  // d_model (K of the projection)
constexpr int ROWS = 56;                // rows kept of the A-panel (S <= 56); fragment rows 56..63 read whatever follows: their outputs are never stored
constexpr int NCW = 192;                // output columns per pass (4 waves x 48)
constexpr int KT = D / 64;              // k-tiles
constexpr int A_BYTES = KT * ROWS * 128;        // 84 KB: [kt][row][64 k] bf16, 16-byte chunks XOR-swizzled by row
constexpr int NSTG = 3;                         // ring of weight tiles: two in flight while one is consumed
constexpr int B_STAGE = NCW * 128;              // 24 KB: [192 rows][64 k]
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ uint32_t lds_off(int row, int kchunk) { return (uint32_t)(row * 128 + ((kchunk ^ (row & 7)) << 4)); }

// B tile of k-tile kt for the weight rows [n0, n0+192): 1536 16-byte slots = 3 per thread
__device__ __forceinline__ void b_dma(const bf16_t* __restrict__ W, int n0, int kt, char* stage, int tid) {
    const int wave_base = tid & ~63;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = tid + i * 512;
        const int row = c >> 3, kc = (c & 7) ^ (row & 7);
        __builtin_amdgcn_global_load_lds(W + (size_t)(n0 + row) * D + kt * 64 + kc * 8, (lds_ptr_t)(stage + (i * 512 + wave_base) * 16), 16, 0, 0);
    }
}

// Variant 2: A-stationary in REGISTERS.  A wave owns 16 rows and keeps their normalised bf16 MFMA fragments for the whole K = 768
// (24 fragments = 96 VGPRs); LDS holds only weight tiles, 384 columns wide (3 stages x 48 KB).  Waves: 4 row blocks x 2 column halves.
constexpr int NCW2 = 384;
constexpr int B2_STAGE = NCW2 * 128;             // 48 KB
__device__ __forceinline__ void b_dma2(const bf16_t* __restrict__ W, int n0, int kt, char* stage, int tid) {
    const int wave_base = tid & ~63;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c = tid + i * 512;
        const int row = c >> 3, kc = (c & 7) ^ (row & 7);
        __builtin_amdgcn_global_load_lds(W + (size_t)(n0 + row) * D + kt * 64 + kc * 8, (lds_ptr_t)(stage + (i * 512 + wave_base) * 16), 16, 0, 0);
    }
}

__global__ __launch_bounds__(512) void ln_qkv_probe_kernel(const float* __restrict__ x, const float* __restrict__ lnw,
                                                           const bf16_t* __restrict__ W, bf16_t* __restrict__ out, float* __restrict__ rstd_out,
                                                           int S, int ncols_per_group, int ldo, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Bs = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, hg = blockIdx.y;
    const int wr = wave & 3, wc = wave >> 2;                       // row block (16 rows), column half (192 of 384)
    const int lrow = lane & 15, lg = lane >> 4;
    const int ncol0 = hg * ncols_per_group;
    const int npass = ncols_per_group / NCW2, ntiles = npass * KT;
    b_dma2(W, ncol0, 0, Bs, tid);
    if (ntiles > 1) b_dma2(W, ncol0 + (1 / KT) * NCW2, 1 % KT, Bs + B2_STAGE, tid);
    // ---- phase 1: this wave's 16 rows, normalised, straight into MFMA A-fragment layout (lane: row lrow, k = 32*f + 8*lg .. +7) ----
    const int row = wr * 16 + lrow;
    const float* xr = x + ((size_t)b * S + (row < S ? row : 0)) * D;
    float ss = 0.f;
#pragma unroll
    for (int f = 0; f < 24; ++f) {
        const float4 p0 = *reinterpret_cast<const float4*>(xr + f * 32 + lg * 8), p1 = *reinterpret_cast<const float4*>(xr + f * 32 + lg * 8 + 4);
        ss += p0.x * p0.x + p0.y * p0.y + p0.z * p0.z + p0.w * p0.w + p1.x * p1.x + p1.y * p1.y + p1.z * p1.z + p1.w * p1.w;
    }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    const float rs = (row < S) ? rsqrtf(ss / (float)D + eps) : 0.f;       // rows beyond the sample: zero operand
    if (lg == 0 && wc == 0 && hg == 0 && row < S && rstd_out) rstd_out[(size_t)b * S + row] = rs;
    bf16x8_t fa[24];
#pragma unroll
    for (int f = 0; f < 24; ++f) {
        const float4 p0 = *reinterpret_cast<const float4*>(xr + f * 32 + lg * 8), p1 = *reinterpret_cast<const float4*>(xr + f * 32 + lg * 8 + 4);
        const float4 w0 = *reinterpret_cast<const float4*>(lnw + f * 32 + lg * 8), w1 = *reinterpret_cast<const float4*>(lnw + f * 32 + lg * 8 + 4);
        union { uint32_t u[4]; bf16x8_t v; } r;
        r.u[0] = pack_bf16x2(w0.x * (p0.x * rs), w0.y * (p0.y * rs)); r.u[1] = pack_bf16x2(w0.z * (p0.z * rs), w0.w * (p0.w * rs));
        r.u[2] = pack_bf16x2(w1.x * (p1.x * rs), w1.y * (p1.y * rs)); r.u[3] = pack_bf16x2(w1.z * (p1.z * rs), w1.w * (p1.w * rs));
        fa[f] = r.v;
    }
    // ---- phase 2: 384 columns per pass; tile t = ps*KT + kt in stage t % 3, tiles t+1, t+2 in flight ----
    int t = 0;
    for (int ps = 0; ps < npass; ++ps) {
        f32x4_t acc[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT; ++kt, ++t) {
            if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + 2 < ntiles) b_dma2(W, ncol0 + ((t + 2) / KT) * NCW2, (t + 2) % KT, Bs + ((t + 2) % 3) * B2_STAGE, tid);
            const char* bt = Bs + (t % 3) * B2_STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t fb[12];
#pragma unroll
                for (int j = 0; j < 12; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(bt + lds_off(wc * 192 + j * 16 + lrow, ks * 4 + lg));
#pragma unroll
                for (int j = 0; j < 12; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[kt * 2 + ks], acc[j], 0, 0, 0);
            }
        }
        if (row < S) {
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int n = ncol0 + ps * NCW2 + wc * 192 + j * 16 + lg * 4;
                uint2 pk;
                pk.x = pack_bf16x2(acc[j][0], acc[j][1]);
                pk.y = pack_bf16x2(acc[j][2], acc[j][3]);
                *reinterpret_cast<uint2*>(out + ((size_t)b * S + row) * ldo + n) = pk;
            }
        }
    }
}

}  // namespace

// x f32 [B*S, 768]; lnw f32 [768]; W bf16 [ngroups*ncols_per_group, 768]; out bf16 [B*S, ldo]; rstd f32 [B*S] or null
extern "C" int xp_ln_qkv(const float* x, const float* lnw, const void* W, void* out, float* rstd, int B, int S, int ngroups,
                         int ncols_per_group, int ldo, float eps, void* stream) {
    if (!x || !lnw || !W || !out || S < 1 || S > 64 || ncols_per_group % NCW2) return 1001;
    const size_t lds = 3 * B2_STAGE;
    static bool set = false;
    if (!set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_qkv_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        set = true;
    }
    hipLaunchKernelGGL(ln_qkv_probe_kernel, dim3(B, ngroups), dim3(512), lds, (hipStream_t)stream, x, lnw, (const bf16_t*)W, (bf16_t*)out,
                       rstd, S, ncols_per_group, ldo, eps);
    return (int)hipGetLastError();
}
