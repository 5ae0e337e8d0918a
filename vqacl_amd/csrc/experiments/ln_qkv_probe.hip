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

__global__ __launch_bounds__(512) void ln_qkv_probe_kernel(const float* __restrict__ x, const float* __restrict__ lnw,
                                                           const bf16_t* __restrict__ W, bf16_t* __restrict__ out, float* __restrict__ rstd_out,
                                                           int S, int ncols_per_group, int ldo, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* A = smem;
    char* Bs = smem + A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ngroups = gridDim.y;
    const int b = blockIdx.x, hg = blockIdx.y;
    (void)ngroups;
    // first weight tile in flight while the rows are normalised
    const int ncol0 = hg * ncols_per_group;
    b_dma(W, ncol0, 0, Bs, tid);
    // ---- phase 1: RMS norm of rows wave, wave+8, ... into the A-panel (bf16, times the norm weight) ----
    float4 wv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) wv[k] = *reinterpret_cast<const float4*>(lnw + lane * 4 + k * 256);
    for (int row = wave; row < ROWS; row += 8) {                 // (rows S..55 are zero filled)
        float4 xv[3];
        float ss = 0.f;
        if (row < S) {
            const float* xr = x + ((size_t)b * S + row) * D;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                xv[k] = *reinterpret_cast<const float4*>(xr + lane * 4 + k * 256);
                ss += xv[k].x * xv[k].x + xv[k].y * xv[k].y + xv[k].z * xv[k].z + xv[k].w * xv[k].w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        ss = wave_sum(ss);
        const float rs = rsqrtf(ss / (float)D + eps);
        if (lane == 0 && row < S && hg == 0 && rstd_out) rstd_out[(size_t)b * S + row] = rs;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int c = lane * 4 + k * 256;                      // column; k-tile c/64, 16-byte chunk (c%64)/8, 4 elements inside it
            uint2 pk;
            pk.x = pack_bf16x2(wv[k].x * (xv[k].x * rs), wv[k].y * (xv[k].y * rs));
            pk.y = pack_bf16x2(wv[k].z * (xv[k].z * rs), wv[k].w * (xv[k].w * rs));
            *reinterpret_cast<uint2*>(A + (c >> 6) * (ROWS * 128) + lds_off(row, (c & 63) >> 3) + (c & 7) * 2) = pk;
        }
    }
    // ---- phase 2: out[b, :, ncol0 + ...] = A-panel x W[ncol0.., :]^T, 192 columns per pass, 12 k-tiles per pass ----
    const int wm = wave >> 2, wn = wave & 3;                       // 2 x 4 waves: wave tile 32 rows x 48 columns
    const int lrow = lane & 15, lg = lane >> 4;
    const int npass = ncols_per_group / NCW;
    // ring: tile t (t = ps * KT + kt) lives in stage t % 3; tiles t+1 and t+2 are in flight while t is consumed
    const int ntiles = npass * KT;
    if (ntiles > 1) b_dma(W, ncol0 + (1 / KT) * NCW, 1 % KT, Bs + 1 * B_STAGE, tid);
    int t = 0;
    for (int ps = 0; ps < npass; ++ps) {
        f32x4_t acc[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < KT; ++kt, ++t) {
            if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");     // tile t landed, tile t+1 may still fly
            else                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + 2 < ntiles) b_dma(W, ncol0 + ((t + 2) / KT) * NCW, (t + 2) % KT, Bs + ((t + 2) % NSTG) * B_STAGE, tid);
            const char* at = A + kt * (ROWS * 128);
            const char* bt = Bs + (t % NSTG) * B_STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t fa[2], fb[3];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(at + lds_off(wm * 32 + i * 16 + lrow, ks * 4 + lg));
#pragma unroll
                for (int j = 0; j < 3; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(bt + lds_off(wn * 48 + j * 16 + lrow, ks * 4 + lg));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
        }
        // epilogue of the pass: lane holds C[m][n..n+3], m = .. + lrow, n = .. + lg*4
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = wm * 32 + i * 16 + lrow;
            if (m >= S) continue;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int n = ncol0 + ps * NCW + wn * 48 + j * 16 + lg * 4;
                uint2 pk;
                pk.x = pack_bf16x2(acc[i][j][0], acc[i][j][1]);
                pk.y = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
                *reinterpret_cast<uint2*>(out + ((size_t)b * S + m) * ldo + n) = pk;
            }
        }
    }
}

}  // namespace

// x f32 [B*S, 768]; lnw f32 [768]; W bf16 [ngroups*ncols_per_group, 768]; out bf16 [B*S, ldo]; rstd f32 [B*S] or null
extern "C" int xp_ln_qkv(const float* x, const float* lnw, const void* W, void* out, float* rstd, int B, int S, int ngroups,
                         int ncols_per_group, int ldo, float eps, void* stream) {
    if (!x || !lnw || !W || !out || S < 1 || S > ROWS || ncols_per_group % NCW) return 1001;
    const size_t lds = A_BYTES + NSTG * B_STAGE;
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
