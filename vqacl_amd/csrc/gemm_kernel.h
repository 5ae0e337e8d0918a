// bf16 MFMA GEMM kernel template for the VL-T5 projections (gfx950, wave64) -- shared by the per-tile translation units
// (gemm_t*.hip: one per tile shape, so that the instantiations compile in parallel) and the dispatcher gemm.hip.
#pragma once
// bf16 MFMA GEMM for the VL-T5 projections (gfx950, wave64).
//
//   C[M,N] = epilogue( alpha * sum_k A[m,k] * B[n,k] )
//
// A is the activation-side operand, B the weight-side operand (nn.Linear keeps W as [N,K]).
// Either operand may be stored "k-major" (element (r,k) at base + k*ld + r) so the three GEMMs of a
// linear layer run on the same kernel without transposed copies:
//   forward  y  = x  W^T : A = x  [M,K] row-major,  B = W  [N,K] row-major
//   dgrad    dx = dy W   : A = dy [M,N] row-major,  B = W  read k-major (k runs over W's rows)
//   wgrad    dW = dy^T x : A = dy read k-major,     B = x  read k-major (k runs over the M rows)
//
// Structure: 4 waves (2x2) over tiles up to 128x128, 8 waves (2x4) over the 256-wide tiles; BK = 64 per step; a ring of 2-3 LDS
// stages filled by LDS-DMA (global_load_lds_dwordx4: as assembly in the kernels with a k-major operand, see lds_dma16) while earlier
// tiles are computed; registers only stage a partial last k-tile.  Row-major operands sit in LDS as [row][64 k] with k contiguous,
// 16-byte slots XOR-swizzled by (row & 7) so the ds_read_b128 fragment reads are bank-conflict free; k-major operands stay
// [64 k][row] in LDS and are gathered into fragments with the hardware transpose read ds_read_b64_tr_b16 (no register transposes).
// MFMA: v_mfma_f32_16x16x32_bf16 with the operands swapped (weight fragment as A, activation
// fragment as B) so each lane ends up with 4 consecutive n of one row m -> 8/16-byte stores.
//
// Fused epilogue (all optional): bias[n], ReLU, gate by the sign of a saved bf16 activation (ReLU and
// dropout backward in one), counter-based dropout, fp32 residual add, accumulate into C, bf16 or
// fp32 output.  Split-K (grid.z) writes fp32 slabs that vlt5_reduce_slabs sums in a fixed order.
#include <cstdlib>
#include <vector>
#include "common.h"
#include "vlt5_hip.h"
#include <hip/hip_ext.h>
#include <type_traits>

// Optional per-workgroup timeline (debug builds only, -DGEMM_TIMELINE): wave 0 of every workgroup records the shader
// clock at fixed points and writes 8 x u64 per workgroup to the buffer registered with vlt5dbg_set_timeline().
#ifdef GEMM_TIMELINE
extern unsigned long long* vlt5_gemm_timeline_buf;      // host side (gemm.hip); travels in GemmArgs (the kernels live in several translation units)
#define g_timeline p.timeline
#define TL_DECL unsigned long long tlv[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define TL(i) do { if (threadIdx.x == 0) { tlv[i] = __builtin_readcyclecounter(); if (i == 0) tlv[7] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define TL_FLUSH() do { if (threadIdx.x == 0 && g_timeline) { \
        tlv[6] = __builtin_amdgcn_s_memrealtime();   /* 100 MHz, common to the whole device: [7] = start, [6] = end */ \
        unsigned long long* o = g_timeline + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8; \
        for (int q = 0; q < 8; ++q) o[q] = tlv[q]; } } while (0)
#else
#define TL_DECL
#define TL(i)
#define TL_FLUSH()
#endif

namespace vlt5gemm {

struct GemmArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    int M, N, K, lda, ldb, ldc;
    float alpha;
    const float* bias;
    const float* resid; int ldr;
    const bf16_t* gate; int ldg; float gate_scale;
    // (ReLU sign bits, vlt5_gemm_desc.gate_bits / relu_bits_out, travel in existing fields so that the argument block -- and with it the
    //  8-wave instantiations' register budget -- does not grow: ldg < 0: `gate` points at the bit matrix, -ldg bytes per row; a ReLU bf16
    //  epilogue with emit_xw != null: emit_xw is the bit matrix to write, ldr bytes per row)
    uint32_t drop_thr, drop_seed;
    int relu, out_f32, accum;
    int ktiles_per_split; long long c_split_stride;
    long long batch_a, batch_b, batch_c;          // element strides between batch entries (blockIdx.z)
    // T5 RMS norm folded around the GEMM (vlt5_gemm_desc.norm_*).  Consumer side: A holds bf16(x * w_norm) and the epilogue scales
    // row m by rstd[m] = rsqrt(sum of its rs_n partial sums of squares / d + eps); tiles of column 0 also store rstd[m].  Producer
    // side (f32 residual epilogue): the finished row values v are also written as bf16(v * emit_w[n]) to emit_xw and each wave
    // leaves sum(v^2) over its columns in emit_ssq[m * SSQ_STRIDE + tile_n * WN + wave_n].
    const float* rs_part; int rs_n; float rs_inv_d, rs_eps; float* rstd_out;
    const float* emit_w; bf16_t* emit_xw; float* emit_ssq;
    bf16_t* C2;                                   // optional bf16 copy of a plain f32 output (same indexing as C), or null
    float* sumsq; long long sumsq_zstride;        // optional: sum of squares of the tile's (plain f32) output -> sumsq[z*zstride + tile]
    // grouped launch (flat grid: x = every tile of every batch entry of problem 1, then those of problem 2; y = z = 1): workgroups
    // [grp_tiles, gridDim.x) work on a SECOND problem with its own shape, reduction length and batch count, same (plain f32) epilogue
    int grp_tiles, grp_t1, grp_t2;                // workgroups of problem 1; tiles per batch entry of problem 1 / 2
    const bf16_t* gA; const bf16_t* gB; void* gC; int gM, gN, gK, glda, gldb, gldc;
    long long gbatch_a, gbatch_b, gbatch_c;
    bf16_t* gC2; float* gsumsq; long long gsumsq_zstride;
#ifdef GEMM_TIMELINE
    unsigned long long* timeline;
#endif
};

constexpr int BK = 64;
constexpr int SSQ_STRIDE = 32;          // floats per row of a sum-of-squares partials buffer (<= 32 column slices of a row)

// Folded norm, consumer side: rstd of a row from the partial sums of squares the producing GEMM's epilogue left.  The partials of
// up to RS_ROWS rows per lane are REQUESTED with norm_request() before the kernel's first DMA requests and reduced with
// norm_reduce() behind them (one memory round trip, hidden under the prologue); a wave's rows are spread over its lanes (row
// q*64 + lane), the epilogue fetches the factor of "its" fragment rows with a lane shuffle.  Fixed summation order.
constexpr int RS_V = 8;                  // float4 per row: SSQ_STRIDE / 4
struct NormRaw { float4 v[RS_V]; };
__device__ __forceinline__ void norm_request(const float* __restrict__ part, int n, int row, NormRaw& r) {
    const float4* q = reinterpret_cast<const float4*>(part + (size_t)row * SSQ_STRIDE);
#pragma unroll
    for (int k = 0; k < RS_V; ++k)
        if (k * 4 < n) r.v[k] = q[k];                         // (n is block-uniform: whole float4s, the tail is masked in the sum)
}
__device__ __forceinline__ float norm_reduce(const NormRaw& r, int n, float inv_d, float eps) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < RS_V; ++k) {
        if (k * 4 < n) {
            s += r.v[k].x;
            s += (k * 4 + 1 < n) ? r.v[k].y : 0.f;
            s += (k * 4 + 2 < n) ? r.v[k].z : 0.f;
            s += (k * 4 + 3 < n) ? r.v[k].w : 0.f;
        }
    }
    return rsqrtf(s * inv_d + eps);
}

__device__ __forceinline__ uint32_t lds_off(int row, int kchunk) {            // byte offset in a [R][64] bf16 tile
    return (uint32_t)(row * 128 + ((kchunk ^ (row & 7)) << 4));
}

// Output stores go through the L2 (common.h store_wt16: the dirty lines of a launch would otherwise be written back when the kernel
// ends, after its last workgroup)
__device__ __forceinline__ void st16(void* dst, uint4 v) { store_wt16(dst, v); }
__device__ __forceinline__ void st16f(float* dst, float4 v) { store_wt16f(dst, v); }

// ---- row-major operand: tile [R rows][64 k], 16-byte chunks along k ------------------------------
template <int R, int NT>
__device__ __forceinline__ void gload_rm(const bf16_t* __restrict__ base, int ld, int row0, int k0, int rmax, int K,
                                         uint4 (&v)[8], int tid) {
#pragma unroll
    for (int i = 0; i < (R * 8 + NT - 1) / NT; ++i) {
        int c = tid + i * NT;
        int row = c >> 3, kc = c & 7;
        int gr = row0 + row, gk = k0 + kc * 8;
        uint4 z = make_uint4(0, 0, 0, 0);
        if (gr < rmax && gk < K) z = *reinterpret_cast<const uint4*>(base + (size_t)gr * ld + gk);
        v[i] = z;
    }
}
template <int R, int NT>
__device__ __forceinline__ void lstore_rm(char* tile, const uint4 (&v)[8], int tid) {
#pragma unroll
    for (int i = 0; i < (R * 8 + NT - 1) / NT; ++i) {
        int c = tid + i * NT;
        int row = c >> 3, kc = c & 7;
        *reinterpret_cast<uint4*>(tile + lds_off(row, kc)) = v[i];
    }
}

// ---- k-major operand: storage [K][R'] (r contiguous).  The tile is kept in LDS exactly as it lies in memory, [64 k][R] with
// r contiguous (same 16-byte global loads / ds_write_b128 as a row-major operand, no VALU work), and the MFMA fragments are
// gathered with the gfx950 transpose read ds_read_b64_tr_b16: within each 16-lane group, lane l receives element (l&3) of
// the 8 bytes addressed by lanes 4j + (l>>2), j = 0..3  (semantics pinned by tests/test_gpu_probe.py).  Lane i of a group
// therefore points at T[k0 + (i>>2)][r0 + 4*(i&3)] and ends up with T[k0 .. k0+3][r0 + i]: four consecutive k of "its" row.
// 16-byte slots are XOR-swizzled per k-row so that the 8 k-rows x 32 bytes touched by one 32-lane half land on 64
// distinct banks.
template <int R>
__device__ __forceinline__ int km_swz(int k) {
    return R >= 128 ? (((k & 3) << 1) | (((k >> 3) & 1) << 3)) : ((((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2));
}
template <int R, int NT>
__device__ __forceinline__ void gload_km(const bf16_t* __restrict__ base, int ld, int row0, int k0, int rmax, int K,
                                         uint4 (&v)[8], int tid) {
    constexpr int CPR = R / 8;                                  // 16-byte chunks per k-row
#pragma unroll
    for (int i = 0; i < R * 8 / NT; ++i) {
        int c = tid + i * NT;
        int k = c / CPR, rc = c % CPR;
        int gk = k0 + k, gr = row0 + rc * 8;
        uint4 z = make_uint4(0, 0, 0, 0);
        if (gr < rmax && gk < K) z = *reinterpret_cast<const uint4*>(base + (size_t)gk * ld + gr);
        v[i] = z;
    }
}
template <int R, int NT>
__device__ __forceinline__ void lstore_km(char* tile, const uint4 (&v)[8], int tid) {
    constexpr int CPR = R / 8;
#pragma unroll
    for (int i = 0; i < R * 8 / NT; ++i) {
        int c = tid + i * NT;
        int k = c / CPR, rc = c % CPR;
        *reinterpret_cast<uint4*>(tile + k * (R * 2) + ((rc ^ km_swz<R>(k)) << 4)) = v[i];
    }
}
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ s16x4_t lds_tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
}
// fragment of rows r0..r0+15, k = ks*32 + 8*(lane>>4) .. +7, from a k-major tile
template <int R>
__device__ __forceinline__ bf16x8_t frag_km(const char* tile, int r0, int ks, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int k = ks * 32 + g * 8 + (i >> 2);
    const int cb = r0 * 2 + (i & 3) * 8;
    const char* a = tile + k * (R * 2) + (((cb >> 4) ^ km_swz<R>(k)) << 4) + (cb & 15);
    s16x4_t lo = lds_tr_read(a);
    s16x4_t hi = lds_tr_read(a + 4 * (R * 2));                  // k + 4: same swizzle
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// ---- direct global -> LDS staging (global_load_lds_dwordx4): no staging VGPRs, no ds_write.  The LDS destination of a
// wave-instruction is linear (wave-uniform base + lane*16), so the XOR swizzle is applied to the per-lane SOURCE address
// inside the same 128-byte row segment (same cache line, coalescing unchanged).  Only for full k-tiles; rows past the edge
// are clamped to a valid row (their products land in outputs the epilogue never stores).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
// One wave-instruction of LDS-DMA: 64 lanes x 16 bytes from the per-lane global addresses to `dst_wave` + lane * 16.
// ASM = true issues it as inline assembly.  Why: the compiler's wait-count pass knows that the builtin writes LDS asynchronously
// and, for the transpose-read builtin (ds_read_b64_tr_b16, every k-major operand), cannot tell the stage being read from the
// stage being filled -- it puts an `s_waitcnt vmcnt(0)` in front of the first transpose read that follows a DMA request.  That
// wait also covers the k-tiles requested one and two steps AHEAD, i.e. it turns the 3-stage ring into "request, then wait for it"
// and puts the whole global-load latency into every k-step of every dgrad / wgrad GEMM (and four times per step into the
// interleaved 8-wave k-step).  Plain ds_read_b128 reads do not get that wait.  Through inline assembly the compiler does not
// model the write; the synchronisation is what the loops spell out anyway (counted `s_waitcnt vmcnt(n)` + `s_barrier` before a
// stage is read, `vmcnt(0)` before LDS is reused).  M0 carries the wave-uniform LDS address; nothing else in these kernels uses M0.
#ifndef GEMM_ASM_DMA
#define GEMM_ASM_DMA 2                  // 0: builtin everywhere, 1: assembly everywhere, 2: assembly in kernels with a k-major operand
#endif
#ifndef GEMM_DMA_NT
#define GEMM_DMA_NT 0                   // 1: nt cache policy on the operand stream (experiment)
#endif
template <bool ASM>
__device__ __forceinline__ void lds_dma16(const bf16_t* src, char* dst_wave) {
    if constexpr (ASM) {
        const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)dst_wave);
#if GEMM_DMA_NT
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(m0v), "v"(src));
#else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(src));
#endif
    } else {
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)dst_wave, 16, 0, GEMM_DMA_NT ? 2 : 0);
    }
}
// Experiment switch (off): L2 prefetch of a k-tile that the ring has no stage for yet (GEMM_L2PF = its distance in k-tiles behind the
// tile being requested).  Question: is the k-step (8-wave tiles 1.2-1.3 us whether 9 or 240 workgroups run, against ~1.0 us of MFMAs;
// 3-stage tiles 0.52 us; profiles/r04_gemm_workgroup_timeline.txt) the latency of the stage's L2 MISSES (an L2 hit returns in ~300
// cycles, TCP_TCC_READ_REQ_LATENCY / _REQ, 60-85 % of the requests hit)?  Each tile request is followed by ONE more LDS-DMA
// instruction per wave that reads one word of every 128-byte line of a LATER tile (BM + BN lines: one per lane) into a scratch strip
// behind the stages; it counts in vmcnt like a piece, issued last, and "tile landed" waits leave it outstanding.  Answer: no --
// same-box A/B (profiles/r04_i_gemm_l2_prefetch.txt) 8.76 ms per step without, 9.45 / 9.65 / 9.90 ms at distance 1 / 2 / 3; the
// warm 256 x 256 k-step stays 1.23 us, the 3-stage tiles go 0.55 -> 0.70 us.  The k-step is issue-side work (the LDS + MFMA step
// alone measures 2357 cycles, tools/mfma_shape_probe.hip), and 64 more cache lines per wave and step are that much more of it.
#ifndef GEMM_L2PF
#define GEMM_L2PF 0
#endif
template <bool ASM>
__device__ __forceinline__ void lds_dma4(const void* src, char* dst_wave) {
    if constexpr (ASM) {
        const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)dst_wave);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(m0v), "v"(src));
    } else {
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)dst_wave, 4, 0, 0);
    }
}
template <int R, int NT, bool ASM = false>
__device__ __forceinline__ void glds_rm(const bf16_t* __restrict__ base, int ld, int row0, int k0, int rmax, char* tile, int tid) {
    const int wave_base = (tid & ~63);
#pragma unroll
    for (int i = 0; i < (R * 8 + NT - 1) / NT; ++i) {       // (a tile height that is no multiple of NT/8 rows: the last piece also
                                                             // fetches clamped rows past the tile into the padding of its LDS region)
        const int c = tid + i * NT;
        const int row = c >> 3, kc = (c & 7) ^ (row & 7);
        const int gr = min(row0 + row, rmax - 1);
        lds_dma16<ASM>(base + (size_t)gr * ld + k0 + kc * 8, tile + (i * NT + wave_base) * 16);
    }
}
template <int R, int NT, bool ASM = false>
__device__ __forceinline__ void glds_km(const bf16_t* __restrict__ base, int ld, int row0, int k0, int rmax, char* tile, int tid) {
    constexpr int CPR = R / 8;
    const int wave_base = (tid & ~63);
#pragma unroll
    for (int i = 0; i < R * 8 / NT; ++i) {
        const int c = tid + i * NT;
        const int k = c / CPR, rc = (c % CPR) ^ km_swz<R>(k);
        const int gr = min(row0 + rc * 8, rmax - 8);
        lds_dma16<ASM>(base + (size_t)(k0 + k) * ld + gr, tile + (i * NT + wave_base) * 16);
    }
}

// piece i (0 .. R*8/NT-1) of a tile = NT consecutive 16-byte slots = one wave-instruction per wave
template <int R, int NT, bool ASM = false>
__device__ __forceinline__ void glds_rm_piece(const bf16_t* __restrict__ base, int ld, int row0, int k0, int rmax, char* tile, int tid, int i) {
    const int wave_base = (tid & ~63);
    const int c = tid + i * NT;
    const int row = c >> 3, kc = (c & 7) ^ (row & 7);
    const int gr = min(row0 + row, rmax - 1);
    lds_dma16<ASM>(base + (size_t)gr * ld + k0 + kc * 8, tile + (i * NT + wave_base) * 16);
}
template <int R, int NT, bool ASM = false>
__device__ __forceinline__ void glds_km_piece(const bf16_t* __restrict__ base, int ld, int row0, int k0, int rmax, char* tile, int tid, int i) {
    constexpr int CPR = R / 8;
    const int wave_base = (tid & ~63);
    const int c = tid + i * NT;
    const int k = c / CPR, rc = (c % CPR) ^ km_swz<R>(k);
    const int gr = min(row0 + rc * 8, rmax - 8);
    lds_dma16<ASM>(base + (size_t)(k0 + k) * ld + gr, tile + (i * NT + wave_base) * 16);
}

// DMA pieces of the interleaved k-step (kstep_big): piece p of LPT is issued after the MFMAs of half-step row
// piece_row(p) in [0, 2*FM) -- spread evenly over both halves, or (EARLY) over the first half only.
template <int FM, int LPT, bool EARLY>
__host__ __device__ constexpr int piece_row(int p) { return EARLY ? p * FM / LPT : p * 2 * FM / LPT; }
template <int FM, int LPT, bool EARLY>
__host__ __device__ constexpr int pieces_in_row(int r) {
    int n = 0;
    for (int p = 0; p < LPT; ++p) n += (piece_row<FM, LPT, EARLY>(p) == r) ? 1 : 0;
    return n;
}
// ks = 1 B fragments fetched after the ks = 0 MFMAs of fragment row i: fragment i itself while FM >= FN (the 8-wave layout),
// an even split of the FN fragments over the FM rows otherwise (4-wave layouts with wave tiles wider than tall)
template <int FM, int FN>
__host__ __device__ constexpr int bfrag_lo(int i) { return FM >= FN ? (i < FN ? i : FN) : i * FN / FM; }
// scheduling pattern of the interleaved k-step (see kstep_big): instruction groups in issue order
template <int FM, int FN, int LPT, int RA, int RB, bool EARLY, int I>
__device__ __forceinline__ void pin_ks0() {                 // ks = 0 MFMAs of fragment row I, then the ks = 1 fragments it frees room for
    __builtin_amdgcn_sched_group_barrier(0x008, FN, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, RA + (bfrag_lo<FM, FN>(I + 1) - bfrag_lo<FM, FN>(I)) * RB, 0);
    constexpr int NP = pieces_in_row<FM, LPT, EARLY>(I);
    if constexpr (NP > 0) __builtin_amdgcn_sched_group_barrier(0x010, NP, 0);
    if constexpr (I + 1 < FM) pin_ks0<FM, FN, LPT, RA, RB, EARLY, I + 1>();
}
template <int FM, int FN, int LPT, bool EARLY, int I>
__device__ __forceinline__ void pin_ks1() {
    __builtin_amdgcn_sched_group_barrier(0x008, FN, 0);
    constexpr int NP = pieces_in_row<FM, LPT, EARLY>(FM + I);
    if constexpr (NP > 0) __builtin_amdgcn_sched_group_barrier(0x010, NP, 0);
    if constexpr (I + 1 < FM) pin_ks1<FM, FN, LPT, EARLY, I + 1>();
}

// WM x WN waves share a BM x BN tile: 2x2 (256 threads) for tiles up to 128x128, 2x4 (512 threads) for 256x256.
template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int NS>
__device__ __forceinline__ void gemm_body(GemmArgs p) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN;                 // wave tile
    constexpr int FM = TM / 16, FN = TN / 16;                 // 16x16 fragments per wave
    constexpr int PA_ = (BM * 8 + NT - 1) / NT, PB_ = (BN * 8 + NT - 1) / NT;
    constexpr int LPT = PA_ + PB_;                            // LDS-DMA wave-instructions ("pieces") per k-tile per wave
    // k-major operands on the 4-wave tiles up to 64x128: the transpose reads are 8-byte-per-lane LDS reads, which need many
    // reads in flight per wave to approach the LDS rate -- fetch the fragments of BOTH 32-wide halves up front (and spread
    // the DMA pieces between the MFMAs).  Measured -20..-40 % on dgrad / wgrad shapes; the same scheme costs 5-10 % on
    // row-major short-K shapes and too many registers at 128x128, so it is applied only here.
#ifndef GEMM_KM_STEP_MAX
#define GEMM_KM_STEP_MAX 192
#endif
    constexpr bool KM_STEP = (AKM || BKM) && (WM * WN == 4) && (BM + BN <= GEMM_KM_STEP_MAX);
    // the 8-wave 256x256 kernel runs the hand-interleaved k-step (kstep_big).  (Tried for 4-wave 128x128 with both operands
    // k-major, the layer-batched weight gradients: 385 -> 650 us, the fragment double buffer pushes it into AGPR spills.)
    constexpr bool BIG_STEP = (WM * WN == 8);
    // -DGEMM_PIPE: the software-pipelined k-step of the fused encoder kernel (enc_attn.hip) for the 8-wave tiles: 0 off, 1 every
    // operand layout, 2 only kernels with a k-major operand
#ifndef GEMM_PIPE
#define GEMM_PIPE 2
#endif
    constexpr bool PIPE_STEP = BIG_STEP && (GEMM_PIPE == 1 || (GEMM_PIPE == 2 && (AKM || BKM)));
    constexpr bool ASM_DMA = (GEMM_ASM_DMA == 1) || (GEMM_ASM_DMA == 2 && (AKM || BKM));     // see lds_dma16()
    // -DGEMM_PINGPONG=1 selects the two-role main loop below instead of the interleaved k-step.  Measured (round 1): correct, but
    // no faster where it compiles without spills (4480x3072x768 with k-major B: 36.6 vs 34.8 us) and the all-row-major / all-k-major
    // instantiations spill (302 / 383 VGPRs) -- four barriers and ~0.6 k cycles of DMA issue per k-step eat what the role split
    // gains.  Kept for A/B runs while the one-wave-per-SIMD schedule is worked on.
#ifndef GEMM_PINGPONG
#define GEMM_PINGPONG 0
#endif
    // (region sizes in whole pieces: 224 x 64 is 3.5 pieces of 512 lanes -> 4, the tail holds clamped duplicate rows nobody reads)
    constexpr int A_BYTES = PA_ * NT * 16, B_BYTES = PB_ * NT * 16;
    static_assert(A_BYTES >= BM * BK * 2 && B_BYTES >= BN * BK * 2, "regions hold the tiles");
    static_assert((BM * 8) % NT == 0 || (!AKM && WM * WN == 8), "partial pieces: row-major A of the 8-wave kernel only");
    static_assert((BN * 8) % NT == 0, "B tiles are whole pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;        // stage s: A tile at s*STAGE_BYTES, B tile right after it
    constexpr int NSTAGE = NS;                            // default: 3 stages up to 64x128 (72 KB, 2 workgroups/CU), 2 for 128x128

    TL_DECL;
    TL(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile mapping: the dispatcher places workgroup b on XCD b % 8 (each XCD has a private 4 MB L2), so give
    // every XCD one contiguous run of tiles, n fastest: the tiles that share an A row-panel run on the same L2 back to
    // back, and the weight panel stays L2-resident per XCD.  Bijective for any tile count (speed only, never correctness).
    int bid = blockIdx.x, zb = blockIdx.z;                   // tile within the batch entry, batch entry
    if (p.grp_tiles > 0) {                                   // (workgroup-uniform) grouped launch: a flat grid over both problems
        int tpb = p.grp_t1;
        if (bid >= p.grp_tiles) {                            // the second problem
            bid -= p.grp_tiles;
            tpb = p.grp_t2;
            p.A = p.gA; p.B = p.gB; p.C = p.gC; p.M = p.gM; p.N = p.gN; p.K = p.gK; p.lda = p.glda; p.ldb = p.gldb; p.ldc = p.gldc;
            p.batch_a = p.gbatch_a; p.batch_b = p.gbatch_b; p.batch_c = p.gbatch_c;
            p.C2 = p.gC2; p.sumsq = p.gsumsq; p.sumsq_zstride = p.gsumsq_zstride;
        }
        zb = bid / tpb;
        bid -= zb * tpb;
    }
    const int gx = (p.N + BN - 1) / BN, gy = (p.M + BM - 1) / BM;
    const int ntiles = gx * gy;
    int tile_id;
    {
        const int b = bid, q = ntiles >> 3, r = ntiles & 7, xcd = b & 7, loc = b >> 3;
        tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int m0 = (tile_id / gx) * BM, n0 = (tile_id % gx) * BN;
    const int nk_total = (p.K + BK - 1) / BK;
    int kt0 = 0, kt1 = nk_total;
    char* Cbase = reinterpret_cast<char*>(p.C) + (long long)zb * p.batch_c * (p.out_f32 ? 4 : 2);
    bf16_t* C2base = p.C2 ? p.C2 + (long long)zb * p.batch_c : nullptr;
    p.A += (long long)zb * p.batch_a;
    p.B += (long long)zb * p.batch_b;
    if (p.resid) p.resid += (long long)zb * p.batch_c;
    if (p.ktiles_per_split > 0) {
        kt0 = blockIdx.y * p.ktiles_per_split;
        kt1 = min(nk_total, kt0 + p.ktiles_per_split);
        Cbase += (size_t)blockIdx.y * (size_t)p.c_split_stride * 4;
    }

    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    uint4 ra[8], rb[8];          // tail k-tile through registers: R*8/NT <= 8 chunks per thread
    auto gload = [&](int kt) __attribute__((always_inline)) {
        if (AKM) gload_km<BM, NT>(p.A, p.lda, m0, kt * BK, p.M, p.K, ra, tid);
        else     gload_rm<BM, NT>(p.A, p.lda, m0, kt * BK, p.M, p.K, ra, tid);
        if (BKM) gload_km<BN, NT>(p.B, p.ldb, n0, kt * BK, p.N, p.K, rb, tid);
        else     gload_rm<BN, NT>(p.B, p.ldb, n0, kt * BK, p.N, p.K, rb, tid);
    };
    auto lstore = [&](int s) __attribute__((always_inline)) {
        char* at = smem + s * STAGE_BYTES;
        char* bt = at + A_BYTES;
        if (AKM) lstore_km<BM, NT>(at, ra, tid); else lstore_rm<BM, NT>(at, ra, tid);
        if (BKM) lstore_km<BN, NT>(bt, rb, tid); else lstore_rm<BN, NT>(bt, rb, tid);
    };

    // L2 prefetch (GEMM_L2PF, see lds_dma4): lane -> one line of the A tile (lanes < BM) or of the B tile; the line of k-tile kt0,
    // advanced by pf_step bytes per k-tile.  k-tiles past the last full one re-request the last (the waits count on one per request).
    static_assert(BM + BN <= NT, "one line per lane");
    constexpr int PF = GEMM_L2PF ? 1 : 0, VPT = LPT + PF;          // vmcnt units of the prefetch / of a whole tile request
    const char* pf_base = nullptr;
    long long pf_step = 0;
    const int pf_last = kt1 - 1 - (((kt1 == nk_total) && (p.K % BK != 0)) ? 1 : 0) - kt0;      // index of the last full k-tile of this workgroup
    if constexpr (PF) {
        const bool isa = tid < BM;
        const int r = isa ? tid : min(tid - BM, BN - 1);
        const bf16_t* q;
        if (isa) {
            if (AKM) { constexpr int LPR = BM / 64 > 0 ? BM / 64 : 1; q = p.A + (size_t)(kt0 * BK + r / LPR) * p.lda + max(min(m0 + (r % LPR) * 64, p.M - 8), 0); pf_step = (long long)BK * p.lda * 2; }
            else     { q = p.A + (size_t)min(m0 + r, p.M - 1) * p.lda + kt0 * BK; pf_step = BK * 2; }
        } else {
            if (BKM) { constexpr int LPR = BN / 64 > 0 ? BN / 64 : 1; q = p.B + (size_t)(kt0 * BK + r / LPR) * p.ldb + max(min(n0 + (r % LPR) * 64, p.N - 8), 0); pf_step = (long long)BK * p.ldb * 2; }
            else     { q = p.B + (size_t)min(n0 + r, p.N - 1) * p.ldb + kt0 * BK; pf_step = BK * 2; }
        }
        pf_base = reinterpret_cast<const char*>(q);
    }
    auto l2pf = [&](int kt) __attribute__((always_inline)) {          // kt: the tile being requested; the prefetch goes GEMM_L2PF tiles behind it
        if constexpr (PF) {
            const int rel = max(min(kt - kt0 + GEMM_L2PF, pf_last), 0);
            lds_dma4<ASM_DMA>(pf_base + (long long)rel * pf_step, smem + NSTAGE * STAGE_BYTES + (tid & ~63) * 4);
        }
    };
    auto glds = [&](int kt, int s) __attribute__((always_inline)) {                        // asynchronous: completion is awaited with vmcnt(0)
        char* at = smem + s * STAGE_BYTES;
        char* bt = at + A_BYTES;
        if (AKM) glds_km<BM, NT, ASM_DMA>(p.A, p.lda, m0, kt * BK, p.M, at, tid); else glds_rm<BM, NT, ASM_DMA>(p.A, p.lda, m0, kt * BK, p.M, at, tid);
        if (BKM) glds_km<BN, NT, ASM_DMA>(p.B, p.ldb, n0, kt * BK, p.N, bt, tid); else glds_rm<BN, NT, ASM_DMA>(p.B, p.ldb, n0, kt * BK, p.N, bt, tid);
        l2pf(kt);
    };
    const int lrow = lane & 15, lg = lane >> 4;
    auto glds_piece = [&](int kt, int s, int pc) __attribute__((always_inline)) {
        char* at = smem + s * STAGE_BYTES;
        char* bt = at + A_BYTES;
        constexpr int PA = PA_;
        if (pc < PA) {
            if (AKM) glds_km_piece<BM, NT, ASM_DMA>(p.A, p.lda, m0, kt * BK, p.M, at, tid, pc); else glds_rm_piece<BM, NT, ASM_DMA>(p.A, p.lda, m0, kt * BK, p.M, at, tid, pc);
        } else {
            if (BKM) glds_km_piece<BN, NT, ASM_DMA>(p.B, p.ldb, n0, kt * BK, p.N, bt, tid, pc - PA);
            else     glds_rm_piece<BN, NT, ASM_DMA>(p.B, p.ldb, n0, kt * BK, p.N, bt, tid, pc - PA);
        }
        if (pc == LPT - 1) l2pf(kt);                       // (the last piece of a tile request)
    };
    auto ldA = [&](const char* at, int i, int ks) __attribute__((always_inline)) -> bf16x8_t {
        const int r0 = wm * TM + i * 16;
        if (AKM) return frag_km<BM>(at, r0, ks, lane);
        return *reinterpret_cast<const bf16x8_t*>(at + lds_off(r0 + lrow, ks * 4 + lg));
    };
    auto ldB = [&](const char* bt, int j, int ks) __attribute__((always_inline)) -> bf16x8_t {
        const int r0 = wn * TN + j * 16;
        if (BKM) return frag_km<BN>(bt, r0, ks, lane);
        return *reinterpret_cast<const bf16x8_t*>(bt + lds_off(r0 + lrow, ks * 4 + lg));
    };
    auto kstep_km = [&](int stage, int kt_pf, int s_pf) __attribute__((always_inline)) {
        const char* at = smem + stage * STAGE_BYTES;
        const char* bt = at + A_BYTES;
        bf16x8_t fa[2][FM], fb[2][FN];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[ks][i] = ldA(at, i, ks);
#pragma unroll
            for (int j = 0; j < FN; ++j) fb[ks][j] = ldB(bt, j, ks);
        }
        constexpr int NMF = 2 * FM * FN;                    // MFMAs of the k-step
#pragma unroll
        for (int q = 0; q < NMF; ++q) {
#pragma unroll
            for (int pc = 0; pc < LPT; ++pc)
                if (pc * NMF / LPT == q) glds_piece(kt_pf, s_pf, pc);      // piece pc goes in front of MFMA number pc*NMF/LPT
            const int ks = q / (FM * FN), i = (q / FN) % FM, j = q % FN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
        }
    };
    // k-step of the 8-wave 256x256 kernel (64 MFMAs, 24 fragment reads, 8 LDS-DMA pieces per wave), finely interleaved: all
    // waves of the workgroup run in lockstep behind the barrier, so whatever a wave issues in a bunch (the 8 DMA pieces cost
    // ~100+ issue cycles each, the fragment reads have ~100 cycles of latency) leaves the matrix pipe of its SIMD idle.
    // Order: ks=0 fragments; then per fragment row FN MFMAs followed by the ks=1 fragment reads whose registers that row
    // frees and a DMA piece of the NEXT k-tile; then the ks=1 MFMAs.
    // Row-major operands: all FM pieces go into the first half, so the ks = 1 MFMAs cover their latency before the next step's
    // vmcnt(0) (-6 % per launch); with a k-major operand (two transpose reads per fragment) spreading them over both halves
    // measured better.
#ifndef GEMM_KM_EARLY
#define GEMM_KM_EARLY 0
#endif
    constexpr bool EARLY = (!AKM && !BKM) || GEMM_KM_EARLY;
    auto kstep_big = [&](int stage, int kt_pf, int s_pf) __attribute__((always_inline)) {
        const char* at = smem + stage * STAGE_BYTES;
        const char* bt = at + A_BYTES;
        bf16x8_t fa0[FM], fb0[FN], fa1[FM], fb1[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) fa0[i] = ldA(at, i, 0);
#pragma unroll
        for (int j = 0; j < FN; ++j) fb0[j] = ldB(bt, j, 0);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j], fa0[i], acc[i][j], 0, 0, 0);
            fa1[i] = ldA(at, i, 1);
#pragma unroll
            for (int j = 0; j < FN; ++j)
                if (j >= bfrag_lo<FM, FN>(i) && j < bfrag_lo<FM, FN>(i + 1)) fb1[j] = ldB(bt, j, 1);
#pragma unroll
            for (int pc = 0; pc < LPT; ++pc)
                if (piece_row<FM, LPT, EARLY>(pc) == i) glds_piece(kt_pf, s_pf, pc);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j], fa1[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int pc = 0; pc < LPT; ++pc)
                if (piece_row<FM, LPT, EARLY>(pc) == FM + i) glds_piece(kt_pf, s_pf, pc);
        }
        constexpr int RA = AKM ? 2 : 1, RB = BKM ? 2 : 1;
        __builtin_amdgcn_sched_group_barrier(0x100, FM * RA + FN * RB, 0);
        pin_ks0<FM, FN, LPT, RA, RB, EARLY, 0>();
        pin_ks1<FM, FN, LPT, EARLY, 0>();
    };
    auto compute = [&](int stage) __attribute__((always_inline)) {
        const char* at = smem + stage * STAGE_BYTES;
        const char* bt = at + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t fa[FM], fb[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int r0 = wm * TM + i * 16;
                if (AKM) fa[i] = frag_km<BM>(at, r0, ks, lane);
                else     fa[i] = *reinterpret_cast<const bf16x8_t*>(at + lds_off(r0 + lrow, ks * 4 + lg));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int r0 = wn * TN + j * 16;
                if (BKM) fb[j] = frag_km<BN>(bt, r0, ks, lane);
                else     fb[j] = *reinterpret_cast<const bf16x8_t*>(bt + lds_off(r0 + lrow, ks * 4 + lg));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    };

    // Main loop over the full k-tiles: NSTAGE-deep ring of LDS stages filled by direct global->LDS loads.  With 3 stages
    // the loads of tile i+2 are issued while tile i is computed and tile i+1 is still in flight: the counted
    // s_waitcnt vmcnt(LPT) at the top of an iteration only waits for the OLDER group, so a whole iteration of MFMAs covers
    // the memory latency.  One raw s_barrier per k-step (it both publishes tile i and retires the reads of tile i-1, whose
    // stage is the one refilled next).  A partial last k-tile (K % 64 != 0) goes through registers with zero fill.
    // GEMM_TOUCH_B (0 off, 1 weights, 2 weights + gate): the weight-side operand of a forward / dgrad GEMM was last read a pass ago and comes from HBM.
    // Every workgroup requests a distinct share of it (one word per 128-byte line) before its own first tile, so the whole
    // panel is on its way into the memory-side cache at once instead of k-tile by k-tile in front of the leading workgroups.
#ifndef GEMM_TOUCH_B
#define GEMM_TOUCH_B 2
#endif
    // The request is a 4-byte LDS-DMA load into a scratch word of LDS behind the stages (one word per lane: launch_one reserves NT * 4
    // bytes) -- NOT a load into a register: a register destination is written asynchronously, long after the asm statement, and the
    // compiler, which does not know that, is free to copy the "value" and hand the register to somebody else.  Rounds 2-5 kept such a
    // register "allocated" by using it at the end of the kernel; in round 6 a second request site made the compiler merge two of them
    // through a copy (and, in the instantiations that spill, spill one), and the late write clobbered an address register: a memory
    // fault.  An LDS destination nobody reads has no such hazard.  Issued as assembly (invisible to the compiler's wait-count pass:
    // the builtin form could make it wait for the request in front of the first fragment read); the main loop's counted waits cover
    // it -- it is the oldest request in the queue.  tools/check_touch_isa.py scans for register-destination requests that remain.
    if constexpr (GEMM_TOUCH_B && !AKM) {
        const int rows_b = BKM ? p.K : p.N, lpr = (BKM ? p.N : p.K) >> 6;          // 64 bf16 = 128 bytes per line
        const long long nlines = (long long)rows_b * lpr;
        const long long nthreads = (long long)gridDim.x * gridDim.y * NT;
        const long long i = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * NT + tid;
        char* const tdst = smem + NSTAGE * STAGE_BYTES + (tid & ~63) * 4;
        auto touch = [&](const void* src) __attribute__((always_inline)) {
            const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)tdst);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(m0v), "v"(src) : "memory");
        };
        if (blockIdx.z == 0 && i < nlines)         // one line of the weight panel per lane (a panel larger than that is only partly requested)
            touch(p.B + (size_t)(i / lpr) * p.ldb + (size_t)(i % lpr) * 64);
#if GEMM_TOUCH_B >= 2
        if (p.gate && p.ldg < 0) {                 // ReLU sign bits ([M][-ldg] bytes): 1/16 of the activation's bytes, one line per lane covers them
            const long long ng = ((long long)p.M * (-p.ldg) + 127) >> 7;
            if (i < ng) touch(reinterpret_cast<const char*>(p.gate) + (size_t)i * 128);
        } else if (p.gate) {                       // the saved bf16 activation the epilogue gates by: as cold as the weights, 6x their size
            const int lprg = p.N >> 6;
            const long long ng = (long long)p.M * lprg, i2 = i + nthreads;
            if (i < ng) touch(p.gate + (size_t)(i / lprg) * p.ldg + (size_t)(i % lprg) * 64);
            if (i2 < ng) touch(p.gate + (size_t)(i2 / lprg) * p.ldg + (size_t)(i2 % lprg) * 64);
        }
#endif
    }
    // folded norm in front of this GEMM: the partial sums of squares of this wave's rows (row q*64 + lane of its TM rows)
    constexpr int RS_ROWS = (TM + 63) / 64;
    constexpr bool NORM_IN = !AKM && !BKM;                   // (forward projections only: both operands row-major)
    NormRaw rs_raw[RS_ROWS];
    float rs_own[RS_ROWS];
#pragma unroll
    for (int q = 0; q < RS_ROWS; ++q) rs_own[q] = 1.f;
    if (NORM_IN && p.rs_part) {                              // block-uniform
#pragma unroll
        for (int q = 0; q < RS_ROWS; ++q) norm_request(p.rs_part, p.rs_n, min(m0 + wm * TM + q * 64 + lane, p.M - 1), rs_raw[q]);
    }
    const bool has_tail = (kt1 == nk_total) && (p.K % BK != 0) && (kt1 > kt0);
    const int nmain = (kt1 - kt0) - (has_tail ? 1 : 0);
    if constexpr (KM_STEP) {
        if (nmain > 0) {
#pragma unroll
            for (int s0 = 0; s0 < NSTAGE - 1; ++s0) glds(min(kt0 + s0, kt0 + nmain - 1), s0);   // always NSTAGE-1 k-tiles in flight
        }
    } else if constexpr (PIPE_STEP) {
        if (nmain > 0) { glds(kt0, 0); glds(min(kt0 + 1, kt0 + nmain - 1), 1); }     // both stages requested
    } else {
#pragma unroll
        for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
            if (s0 < nmain) glds(kt0 + s0, s0);
    }
    if (NORM_IN && p.rs_part) {
#pragma unroll
        for (int q = 0; q < RS_ROWS; ++q) {
            rs_own[q] = norm_reduce(rs_raw[q], p.rs_n, p.rs_inv_d, p.rs_eps);
            const int r = q * 64 + lane, m = m0 + wm * TM + r;
            if (p.rstd_out && n0 == 0 && wn == 0 && r < TM && m < p.M && blockIdx.y == 0) p.rstd_out[m] = rs_own[q];
        }
    }
    int stage = 0, fill = NSTAGE - 1;                      // stage of tile i, stage that tile i+NSTAGE-1 goes to
    TL(1);
    if constexpr (KM_STEP) {
        // branch-free steps (see the 8-wave loop below): the last NSTAGE-1 steps re-request the final k-tile, so the counted
        // vmcnt is a constant and the whole step is one scheduling region
        for (int i = 0; i < nmain; ++i) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF + VPT * (NSTAGE - 2)) : "memory");     // tile i has landed (its prefetch and the later requests may be out)
            __builtin_amdgcn_s_barrier();
#ifdef GEMM_TIMELINE
            if (i == 0) TL(2);
#endif
            kstep_km(stage, min(kt0 + i + NSTAGE - 1, kt0 + nmain - 1), fill);
            stage = (stage + 1 == NSTAGE) ? 0 : stage + 1;
            fill = (fill + 1 == NSTAGE) ? 0 : fill + 1;
        }
        if (nmain > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // surplus prefetches land before LDS is reused / the wave ends
        if (has_tail) __syncthreads();
    } else if constexpr (BIG_STEP && GEMM_PINGPONG) {
        // 8-wave kernel, two roles.  Waves w and w+4 share a SIMD (a workgroup's waves go to the SIMDs cyclically): behind a common
        // barrier they would do the same thing at the same time -- both fetching fragments while the matrix pipe idles, then both
        // issuing MFMAs.  Here the lower four waves ("X") and the upper four ("Y") run half a phase apart; a phase p is one 32-wide
        // half (ks = p & 1) of k-tile t = p >> 1, i.e. FM + FN fragments and FM * FN MFMAs per wave:
        //   segment 1 of phase p:  X: MFMAs of phase p            Y: fragments of phase p   <- stage t & 1
        //   segment 2 of phase p:  X: fragments of phase p + 1    Y: MFMAs of phase p
        // so every SIMD always has one wave on the matrix pipe and one on the LDS / DMA path, and a wave holds the fragments of
        // ONE half k-tile (48 registers) instead of two.  Stage t & 1 is last read in segment 1 of phase 2t+1; tile t+2 is
        // requested into it in the two segments that follow (half of a wave's DMA pieces in each, whatever the wave is doing
        // there) and is awaited by every wave before the barrier that closes segment 1 of phase 2t+3 -- X reads it right after.
        // The last steps re-request the final k-tile into the retired stage instead of branching around the loads.
        static_assert(NSTAGE == 2, "two stages");
        static_assert(LPT % 2 == 0, "the DMA pieces of a wave are issued in two halves");
        const int role = wave >> 2;
        if (nmain > 0) {
            glds(min(kt0 + 1, kt0 + nmain - 1), 1);             // (tile 0 -> stage 0 was requested above)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            TL(2);
            bf16x8_t fa[FM], fb[FN];
            auto load_frags = [&](int ph) __attribute__((always_inline)) {
                const char* at = smem + ((ph >> 1) & 1) * STAGE_BYTES;
                const char* bt = at + A_BYTES;
                const int ks = ph & 1;
#pragma unroll
                for (int i = 0; i < FM; ++i) fa[i] = ldA(at, i, ks);
#pragma unroll
                for (int j = 0; j < FN; ++j) fb[j] = ldB(bt, j, ks);
            };
            // MFMAs of the phase held in fa / fb; HALF >= 0: DMA pieces [HALF*LPT/2, (HALF+1)*LPT/2) of k-tile kt -> stage sdst in between
            auto mfmas = [&](int half, int kt, int sdst) __attribute__((always_inline)) {
                constexpr int NMF = FM * FN, HP = LPT / 2;
#pragma unroll
                for (int q = 0; q < NMF; ++q) {
                    if (half >= 0) {
#pragma unroll
                        for (int pc = 0; pc < HP; ++pc)
                            if (pc * NMF / HP == q) glds_piece(kt, sdst, half * HP + pc);
                    }
                    const int i = q / FN, j = q % FN;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                }
            };
            auto pieces = [&](int half, int kt, int sdst) __attribute__((always_inline)) {
                constexpr int HP = LPT / 2;
#pragma unroll
                for (int pc = 0; pc < HP; ++pc) glds_piece(kt, sdst, half * HP + pc);
            };
            const int last = kt0 + nmain - 1;
            const int P = 2 * nmain;
            if (role == 0) load_frags(0);
            for (int ph = 0; ph < P; ++ph) {
                const int t = ph >> 1, odd = ph & 1;
                // even phase (ph >= 2), segment 1: second half of the pieces of tile t+1 -> stage (t+1)&1 (free since phase 2t-1)
                // odd phase, segment 2: first half of the pieces of tile t+2 -> stage t&1
                if (role == 0) {
                    if (!odd && ph >= 2) mfmas(1, min(kt0 + t + 1, last), (t + 1) & 1); else mfmas(-1, 0, 0);
                    if (odd) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (odd) pieces(0, min(kt0 + t + 2, last), t & 1);
                    load_frags(ph + 1);
                    __builtin_amdgcn_s_barrier();
                } else {
                    if (!odd && ph >= 2) pieces(1, min(kt0 + t + 1, last), (t + 1) & 1);
                    load_frags(ph);
                    if (odd) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (odd) mfmas(0, min(kt0 + t + 2, last), t & 1); else mfmas(-1, 0, 0);
                    __builtin_amdgcn_s_barrier();
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // surplus prefetches land before LDS is reused / the wave ends
        }
        if (has_tail) __syncthreads();
    } else if constexpr (PIPE_STEP) {
        // 8-wave kernel, software-pipelined across the k-steps (the loop of the fused encoder kernel, enc_attn.hip): a k-step is two
        // halves (the 32-deep MFMA slabs ks = 0 / 1 of the 64-deep tile) with the workgroup barrier BETWEEN them --
        //   ks = 0 half of tile t: MFMAs on fragments fetched during the previous step; fetch the ks = 1 fragments of tile t
        //   wait (tile t+1 landed, the ks = 1 fragments returned) + barrier: stage of tile t is free, stage of tile t+1 complete
        //   ks = 1 half of tile t: MFMAs; request tile t+2 into the freed stage; fetch the ks = 0 fragments of tile t+1
        // so no half starts behind a fragment read (with k-major operands the up-front fetch of the barrier-at-the-top step is 24
        // transpose reads, ~700 idle cycles of the matrix pipe per step), and a DMA piece has a whole step to land.
        static_assert(NSTAGE == 2, "two stages");
        if (nmain > 0) {
            bf16x8_t fa0[FM], fb0[FN], fa1[FM], fb1[FN];
            auto half = [&](bf16x8_t (&fa)[FM], bf16x8_t (&fb)[FN], bf16x8_t (&na)[FM], bf16x8_t (&nb)[FN], const char* nat,
                            const char* nbt, int nks, int kt_dma, int s_dma, bool dma) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < FM; ++i) {
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    na[i] = ldA(nat, i, nks);
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        if (j >= bfrag_lo<FM, FN>(i) && j < bfrag_lo<FM, FN>(i + 1)) nb[j] = ldB(nbt, j, nks);
                    if (dma) {
#pragma unroll
                        for (int pc = 0; pc < LPT; ++pc)
                            if (piece_row<FM, LPT, true>(pc) == i) glds_piece(kt_dma, s_dma, pc);
                    }
                }
            };
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF + VPT) : "memory");          // tile 0 (the older group) has landed
            __builtin_amdgcn_s_barrier();
            TL(2);
#pragma unroll
            for (int i = 0; i < FM; ++i) fa0[i] = ldA(smem, i, 0);
#pragma unroll
            for (int j = 0; j < FN; ++j) fb0[j] = ldB(smem + A_BYTES, j, 0);
            int stg = 0;
            for (int i = 0; i < nmain; ++i) {
                const char* at = smem + stg * STAGE_BYTES;
                const char* at2 = smem + (stg ^ 1) * STAGE_BYTES;
                half(fa0, fb0, fa1, fb1, at, at + A_BYTES, 1, 0, 0, false);
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PF) : "memory");
                __builtin_amdgcn_s_barrier();
                half(fa1, fb1, fa0, fb0, at2, at2 + A_BYTES, 0, min(kt0 + i + 2, kt0 + nmain - 1), stg, true);   // (past the end: the last tile again)
                stg ^= 1;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // the surplus prefetch / fragments: before LDS is reused or the wave ends
            stage = 0;
        }
        if (has_tail) __syncthreads();
    } else if constexpr (BIG_STEP) {
        // 8-wave kernel: two stages, branch-free steps.  Every step prefetches; the last one re-requests the final k-tile into
        // the stage that was just retired (harmless, L2-resident) instead of branching around the loads, which keeps the
        // whole step one scheduling region.
        static_assert(NSTAGE == 2, "two stages: the step waits for the single k-tile in flight");
        for (int i = 0; i < nmain; ++i) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF) : "memory");
            __builtin_amdgcn_s_barrier();
#ifdef GEMM_TIMELINE
            if (i == 0) TL(2);
#endif
            kstep_big(stage, min(kt0 + i + 1, kt0 + nmain - 1), stage ^ 1);
            stage ^= 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus prefetch must land before LDS is reused / the wave ends
        if (has_tail) __syncthreads();
    } else
    for (int i = 0; i < nmain; ++i) {
        if (NSTAGE == 3 && i + 1 < nmain) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF + VPT) : "memory");
        else                               asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF) : "memory");
        __builtin_amdgcn_s_barrier();
#ifdef GEMM_TIMELINE
        if (i == 0) TL(2);
#endif
        if (i + NSTAGE - 1 < nmain) glds(kt0 + i + NSTAGE - 1, fill);
        compute(stage);
        stage = (stage + 1 == NSTAGE) ? 0 : stage + 1;
        fill = (fill + 1 == NSTAGE) ? 0 : fill + 1;
    }
    if (has_tail) {
        gload(kt1 - 1);
        lstore(stage);                                     // this stage was last read >= 2 barriers ago
        __syncthreads();
        compute(stage);
    }

    TL(3);
    // ---- epilogue: lane holds C[m][n..n+3], m = .. + (lane&15), n = .. + (lane>>4)*4 -----------
    // Two phases.  (1) ALL auxiliary operands of the wave's tile (residual / C-for-accumulate, or the gate, and the bias) are
    // requested up front; (2) after a single wait every fragment is finished and stored back to back.  A per-fragment
    // "load aux -> wait -> store" sequence would put an s_waitcnt vmcnt(0) between consecutive stores, and on CDNA vmcnt also
    // counts stores: every fragment would wait for the previous fragment's store round trip (measured: 7-20 us of fixed
    // cost per launch before this change).
    const float dscale = drop_scale(p.drop_thr);
    const bool aux_f32 = (p.resid != nullptr) || p.accum;             // block-uniform
    // per-row factor of a bf16 output: alpha, times rstd[m] when a T5 RMS norm is folded in front of this GEMM (GemmArgs.rs_part;
    // bf16 outputs without auxiliary operand only: the q|k|v / query projections and the FFN input projection)
    auto row_factors = [&](float (&rf)[FM]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FM; ++i) rf[i] = p.alpha;
        if (NORM_IN && p.rs_part) {                                       // block-uniform
#pragma unroll
            for (int i = 0; i < FM; ++i) rf[i] *= __shfl(rs_own[(i * 16) >> 6], (i * 16 + lrow) & 63, 64);     // fragment row i*16 + lrow of the wave
        }
    };
    // bf16 output: two neighbouring fragments j, j+1 of a row block are packed and exchanged between the lane rows with
    // v_permlane16_swap (rows 1/3 of fragment j <-> rows 0/2 of fragment j+1), after which a lane owns 8 consecutive columns:
    // one 16-byte store per lane, 64 contiguous bytes per matrix row per instruction, half as many store instructions as
    // 8-byte stores (the epilogue is store-ISSUE bound: ~50 cycles per wave-store whatever its width).  Every lane takes part
    // in the swap; only the store is predicated.
    auto store_pair_bf16_to = [&](bf16_t* base, int m, int j, const float (&v)[4], const float (&w)[4]) __attribute__((always_inline)) {
        const uint32_t p0 = pack_bf16x2(v[0], v[1]), p1 = pack_bf16x2(v[2], v[3]);
        const uint32_t q0 = pack_bf16x2(w[0], w[1]), q1 = pack_bf16x2(w[2], w[3]);
        const auto s0 = __builtin_amdgcn_permlane16_swap(p0, q0, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(p1, q1, false, false);
        const int n = n0 + wn * TN + (j + (lg & 1)) * 16 + (lg >> 1) * 8;
        if (m < p.M && n < p.N)
            st16(base + (size_t)m * p.ldc + n, make_uint4(s0[0], s1[0], s0[1], s1[1]));
    };
    auto store_pair_bf16 = [&](int m, int j, const float (&v)[4], const float (&w)[4]) __attribute__((always_inline)) {
        store_pair_bf16_to(reinterpret_cast<bf16_t*>(Cbase), m, j, v, w);
    };
    // ... and the sign bits of the eight values the lane stores (GemmArgs.bits_out): after the swap a lane owns 8 consecutive columns
    // starting at a multiple of 8 = exactly one byte of the [M][N/8] bit matrix
    auto store_pair_bf16_bits = [&](int m, int j, const float (&v)[4], const float (&w)[4]) __attribute__((always_inline)) {
        const uint32_t p0 = pack_bf16x2(v[0], v[1]), p1 = pack_bf16x2(v[2], v[3]);
        const uint32_t q0 = pack_bf16x2(w[0], w[1]), q1 = pack_bf16x2(w[2], w[3]);
        const auto s0 = __builtin_amdgcn_permlane16_swap(p0, q0, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(p1, q1, false, false);
        const int n = n0 + wn * TN + (j + (lg & 1)) * 16 + (lg >> 1) * 8;
        if (m < p.M && n < p.N) {
            const uint32_t w0 = s0[0], w1 = s1[0], w2 = s0[1], w3 = s1[1];
            st16(reinterpret_cast<bf16_t*>(Cbase) + (size_t)m * p.ldc + n, make_uint4(w0, w1, w2, w3));
            auto nz = [](uint32_t x) { return ((x & 0x7fffu) ? 1u : 0u) | ((x & 0x7fff0000u) ? 2u : 0u); };
            reinterpret_cast<uint8_t*>(p.emit_xw)[(size_t)m * p.ldr + (n >> 3)] = (uint8_t)(nz(w0) | (nz(w1) << 2) | (nz(w2) << 4) | (nz(w3) << 6));
        }
    };
    if (!p.bias && !p.relu && !p.gate && !p.drop_thr && !aux_f32) {
        // plain epilogue (QKV / cross-K/V / lm_head projections, every dgrad without gate, every weight gradient): straight-line
        // scale + pack + store; keeps ~100 option-testing instructions per fragment off the tail of ~70 % of the launches
        const int nb = n0 + wn * TN + lg * 4;
        if (p.out_f32) {
            float ss = 0.f;                       // (sumsq: the gradient-norm share of this tile, see the end of the branch)
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + wm * TM + i * 16 + lrow;
                if (m >= p.M) continue;
                float* crow = reinterpret_cast<float*>(Cbase) + (size_t)m * p.ldc + nb;
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    if (nb + j * 16 < p.N) {
                        const float4 o = make_float4(acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha,
                                                     acc[i][j][3] * p.alpha);
                        ss += o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
                        st16f(crow + j * 16, o);
                        if (C2base) {          // the same values rounded to bf16: a data-parallel bucket's staging copy, no cast pass
                            uint2 pk;
                            pk.x = pack_bf16x2(o.x, o.y);
                            pk.y = pack_bf16x2(o.z, o.w);
                            *reinterpret_cast<uint2*>(C2base + (size_t)m * p.ldc + nb + j * 16) = pk;
                        }
                    }
            }
            // Weight gradients: the clip-by-global-norm of the optimizer needs sum(g^2) over every gradient -- a second pass over
            // 0.9 GB (0.16 ms per step at the HBM roofline).  The tile's share is reduced here in a fixed order (lanes, then waves) and
            // written to the tile's own slot; the optimizer sums the slots in a fixed order too (vlt5_gnorm_finish): deterministic.
            if (p.sumsq && p.ktiles_per_split == 0) {
                ss = wave_sum(ss);
                __syncthreads();                                  // every wave has left the main loop: the stages are free
                float* red = reinterpret_cast<float*>(smem);
                if (lane == 0) red[wave] = ss;
                __syncthreads();
                if (tid == 0) {
                    float t = 0.f;
#pragma unroll
                    for (int w8 = 0; w8 < WM * WN; ++w8) t += red[w8];
                    p.sumsq[(long long)zb * p.sumsq_zstride + tile_id] = t;
                }
            }
        } else {
            float rowf[FM];
            row_factors(rowf);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + wm * TM + i * 16 + lrow;
#pragma unroll
                for (int j = 0; j < FN; j += 2) {
                    float v[4], w[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = acc[i][j][r] * rowf[i]; w[r] = acc[i][j + 1][r] * rowf[i]; }
                    store_pair_bf16(m, j, v, w);
                }
            }
        }
        TL(4);
#ifdef GEMM_TIMELINE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        TL(5);
        TL_FLUSH();
        return;
    }
    // Fused epilogues.  The option set is block-uniform; the combinations the engine issues are compiled as straight-line
    // specialisations (generic lambda + integral constants), everything else takes the fully generic instance.
    const bool drop = p.drop_thr != 0, gate = p.gate != nullptr, bias = p.bias != nullptr, relu = p.relu != 0, f32 = p.out_f32 != 0;
    auto run = [&](auto c_bias, auto c_relu, auto c_gate, auto c_drop, auto c_aux, auto c_f32, auto c_generic) __attribute__((always_inline)) {
        constexpr bool kBias = decltype(c_bias)::value, kRelu = decltype(c_relu)::value, kGate = decltype(c_gate)::value;
        constexpr bool kDrop = decltype(c_drop)::value, kAux = decltype(c_aux)::value, kF32 = decltype(c_f32)::value;
        constexpr bool kGen = decltype(c_generic)::value;      // generic instance: a compiled-in option is still tested at run time
        constexpr bool kRowScale = !kF32 && !kAux && !kGate;   // (the FFN input projection behind a folded norm)
        float rowf[kRowScale ? FM : 1];
        if constexpr (kRowScale) row_factors(rowf);
        const bool do_bias = kBias && (!kGen || bias), do_relu = kRelu && (!kGen || relu), do_drop = kDrop && (!kGen || drop);
        const bool do_aux = kAux && (!kGen || aux_f32);
        float4 bs[FN];
        if constexpr (kBias) {
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = n0 + wn * TN + j * 16 + lg * 4;
                bs[j] = (do_bias && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        constexpr int IC = (FM * FN > 16) ? (16 / FN) : FM;     // fragment rows per pass: at most 16 auxiliary float4 in flight
#pragma unroll
        for (int ib = 0; ib < FM; ib += IC) {
            float4 aux[IC][FN];
            if constexpr (kAux || kGate) {
#pragma unroll
                for (int ii = 0; ii < IC; ++ii) {
                    if (ib + ii >= FM) continue;
                    const int m = m0 + wm * TM + (ib + ii) * 16 + lrow;
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        const int n = n0 + wn * TN + j * 16 + lg * 4;
                        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (m < p.M && n < p.N && (do_aux || kGate)) {
                            if constexpr (kAux) {
                                if (p.resid) t = *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.ldr + n);
                                if (p.accum) {
                                    float4 q = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(Cbase) + (size_t)m * p.ldc + n);
                                    t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
                                }
                            } else if (p.ldg < 0) {          // (kernel-uniform) sign bits: the wave tile's TN bits of row m in ONE load per row
                                if (j == 0) {                    // block (fragment 0 carries them): a quarter of the per-fragment requests
                                    const uint8_t* q = reinterpret_cast<const uint8_t*>(p.gate) + (size_t)m * (size_t)(-p.ldg) + ((n0 + wn * TN) >> 3);
                                    if constexpr (TN == 64) {
                                        if (n0 + wn * TN + 64 <= p.N) {
                                            const uint2 b2 = *reinterpret_cast<const uint2*>(q);
                                            t.x = __uint_as_float(b2.x); t.y = __uint_as_float(b2.y);
                                        } else {                 // (a ragged last column tile: byte by byte, zero beyond N)
                                            uint32_t lo = 0, hi = 0;
                                            for (int b = 0; b < 8; ++b)
                                                if (n0 + wn * TN + b * 8 < p.N) { if (b < 4) lo |= (uint32_t)q[b] << (8 * b); else hi |= (uint32_t)q[b] << (8 * (b - 4)); }
                                            t.x = __uint_as_float(lo); t.y = __uint_as_float(hi);
                                        }
                                    } else {
                                        uint32_t lo = 0;
                                        for (int b = 0; b < TN / 8; ++b)
                                            if (n0 + wn * TN + b * 8 < p.N) lo |= (uint32_t)q[b] << (8 * b);
                                        t.x = __uint_as_float(lo);
                                    }
                                }
                            } else {
                                uint2 g2 = *reinterpret_cast<const uint2*>(p.gate + (size_t)m * p.ldg + n);
                                t.x = __uint_as_float(g2.x); t.y = __uint_as_float(g2.y);
                            }
                        }
                        aux[ii][j] = t;
                    }
                }
            }
            float4 gw[FN];                           // norm-emitting epilogue: the next norm's weight for this lane's columns
            if constexpr (kAux && !kGen && !kBias && !kRelu && !kGate && kF32 && WM * WN == 4 && !AKM && !BKM) {
                if (p.emit_xw) {
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        const int n = n0 + wn * TN + j * 16 + lg * 4;
                        gw[j] = n < p.N ? *reinterpret_cast<const float4*>(p.emit_w + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
            // one explicit vmcnt(0) that EVERY path passes: the loads above sit in divergent branches, and a conservative
            // re-wait before each fragment would land between the stores (vmcnt counts stores too on CDNA)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            auto finish = [&](int ii, int j, int m, float (&v)[4]) __attribute__((always_inline)) {     // everything between the accumulator and the store
                const int i = ib + ii;
                const int n = n0 + wn * TN + j * 16 + lg * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * (kRowScale ? rowf[kRowScale ? i : 0] : p.alpha);
                if constexpr (kBias) { v[0] += bs[j].x; v[1] += bs[j].y; v[2] += bs[j].z; v[3] += bs[j].w; }
                if constexpr (kRelu) {
                    if (do_relu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                }
                if constexpr (kGate) {
                    if (p.ldg < 0) {
                        // byte j*2 + (lg >> 1) of the row block's bits (fragment 0's aux), nibble lg & 1
                        const uint32_t word = __float_as_uint((j * 2 >= 4) ? aux[ii][0].y : aux[ii][0].x);
                        const uint32_t nib = word >> ((((j * 2) & 3) + (lg >> 1)) * 8 + (lg & 1) * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = ((nib >> r) & 1u) ? v[r] * p.gate_scale : 0.f;
                    } else {
                        uint32_t gw[2] = {__float_as_uint(aux[ii][j].x), __float_as_uint(aux[ii][j].y)};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            bf16_t h = (bf16_t)((gw[r >> 1] >> ((r & 1) * 16)) & 0xffffu);
                            v[r] = (bf16_to_f32(h) > 0.f) ? v[r] * p.gate_scale : 0.f;
                        }
                    }
                }
                if constexpr (kDrop) {
                    if (do_drop) {
                        bool kp[4];
                        drop_keep4(p.drop_seed, (uint32_t)m * (uint32_t)p.N + (uint32_t)n, p.drop_thr, kp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = kp[r] ? v[r] * dscale : 0.f;
                    }
                }
                if constexpr (kAux) { v[0] += aux[ii][j].x; v[1] += aux[ii][j].y; v[2] += aux[ii][j].z; v[3] += aux[ii][j].w; }
            };
#pragma unroll
            for (int ii = 0; ii < IC; ++ii) {
                if (ib + ii >= FM) continue;
                const int m = m0 + wm * TM + (ib + ii) * 16 + lrow;
                if constexpr (kF32) {
                    if constexpr (kAux && !kGen && !kBias && !kRelu && !kGate && WM * WN == 4 && !AKM && !BKM) {      // (the dispatcher gives a norm-emitting GEMM a 4-wave tile)
                        if (p.emit_xw) {          // (block-uniform) the next sublayer's T5 RMS norm is folded around its GEMM: see GemmArgs
                            float ss = 0.f;
#pragma unroll
                            for (int j = 0; j < FN; j += 2) {
                                float v[4], w[4], ev[4], ew[4];
                                finish(ii, j, m, v);
                                finish(ii, j + 1, m, w);
                                const int n = n0 + wn * TN + j * 16 + lg * 4;
                                const bool in0 = m < p.M && n < p.N, in1 = m < p.M && n + 16 < p.N;
                                const float4 g0 = gw[j], g1 = gw[j + 1];
                                if (in0) {
                                    st16f(reinterpret_cast<float*>(Cbase) + (size_t)m * p.ldc + n, make_float4(v[0], v[1], v[2], v[3]));
                                    ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                                }
                                if (in1) {
                                    st16f(reinterpret_cast<float*>(Cbase) + (size_t)m * p.ldc + n + 16, make_float4(w[0], w[1], w[2], w[3]));
                                    ss += w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + w[3] * w[3];
                                }
                                ev[0] = v[0] * g0.x; ev[1] = v[1] * g0.y; ev[2] = v[2] * g0.z; ev[3] = v[3] * g0.w;
                                ew[0] = w[0] * g1.x; ew[1] = w[1] * g1.y; ew[2] = w[2] * g1.z; ew[3] = w[3] * g1.w;
                                store_pair_bf16_to(p.emit_xw, m, j, ev, ew);
                            }
                            ss += __shfl_xor(ss, 16, 64);
                            ss += __shfl_xor(ss, 32, 64);
                            if (lg == 0 && m < p.M) p.emit_ssq[(size_t)m * SSQ_STRIDE + (n0 / BN) * WN + wn] = ss;
                            continue;
                        }
                    }
                    if (m >= p.M) continue;
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        const int n = n0 + wn * TN + j * 16 + lg * 4;
                        if (n >= p.N) continue;
                        float v[4];
                        finish(ii, j, m, v);
                        float* c = reinterpret_cast<float*>(Cbase) + (size_t)m * p.ldc + n;
                        st16f(c, make_float4(v[0], v[1], v[2], v[3]));
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < FN; j += 2) {
                        float v[4], w[4];
                        finish(ii, j, m, v);
                        finish(ii, j + 1, m, w);
                        if constexpr (kRelu && !kGen) {
                            if (p.emit_xw) { store_pair_bf16_bits(m, j, v, w); continue; }      // (kernel-uniform: the bit matrix)
                        }
                        store_pair_bf16(m, j, v, w);
                    }
                }
            }
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    if (aux_f32 && !gate && !bias && !relu && f32) {                 // residual GEMMs (attention O, FFN wo), accumulate
        if (drop) run(F{}, F{}, F{}, T{}, T{}, T{}, F{}); else run(F{}, F{}, F{}, F{}, T{}, T{}, F{});
    } else if (relu && !aux_f32 && !gate && !bias && !f32) {         // FFN wi: ReLU (+dropout) -> bf16
        if (drop) run(F{}, T{}, F{}, T{}, F{}, F{}, F{}); else run(F{}, T{}, F{}, F{}, F{}, F{}, F{});
    } else if (gate && !aux_f32 && !bias && !relu && !drop && !f32) {   // FFN hidden gradient, gated by the saved activation
        run(F{}, F{}, T{}, F{}, F{}, F{}, F{});
    } else if (bias && !aux_f32 && !gate && !relu && !drop && f32) {    // visual projection
        run(T{}, F{}, F{}, F{}, F{}, T{}, F{});
    } else if (gate) {                                                // anything else: options tested at run time
        if (f32) run(T{}, T{}, T{}, T{}, F{}, T{}, T{}); else run(T{}, T{}, T{}, T{}, F{}, F{}, T{});
    } else {
        if (f32) run(T{}, T{}, F{}, T{}, T{}, T{}, T{}); else run(T{}, T{}, F{}, T{}, T{}, F{}, T{});
    }
    TL(4);
#ifdef GEMM_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    TL(5);
    TL_FLUSH();
}

template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int NS>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(GemmArgs p) {
    gemm_body<BM, BN, WM, WN, AKM, BKM, NS>(p);
}
static __global__ void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out, long long n,
                                    int nslabs, long long stride, int accum, bf16_t* __restrict__ out16) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    float4 s = accum ? *reinterpret_cast<const float4*>(out + i) : make_float4(0, 0, 0, 0);
    for (int k = 0; k < nslabs; ++k) {
        float4 q = *reinterpret_cast<const float4*>(slabs + (size_t)k * stride + i);
        s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
    }
    *reinterpret_cast<float4*>(out + i) = s;
    if (out16) {
        uint2 pk;
        pk.x = pack_bf16x2(s.x, s.y);
        pk.y = pack_bf16x2(s.z, s.w);
        *reinterpret_cast<uint2*>(out16 + i) = pk;
    }
}

// ---- measurement hook (vlt5_gemm_timing_*): while enabled, every gemm_kernel dispatch carries a start / stop event of its own
// (hipExtLaunchKernelGGL attaches them to the dispatch, so the elapsed time is the kernel's execution time as a profiler sees it,
// not the distance between two stream markers).  Process-global, measurement only, not thread-safe.
struct TimingState {
    bool on = false;
    std::vector<hipEvent_t> ev;               // 2 per record
    std::vector<vlt5_gemm_timing_rec> rec;
    size_t cap = 0;
};
}  // namespace
extern vlt5gemm::TimingState vlt5_gemm_timing_state;      // defined in gemm.hip
namespace vlt5gemm {
#define g_timing vlt5_gemm_timing_state

template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int NS>
int launch_one(const GemmArgs& a, dim3 grid, hipStream_t st) {
    constexpr int NT_ = WM * WN * 64;
    constexpr size_t lds = (size_t)NS * ((BM * 8 + NT_ - 1) / NT_ + (BN * 8 + NT_ - 1) / NT_) * NT_ * 16 + NT_ * 4;     // + one scratch word per lane: the destination of the panel-touch requests (and of GEMM_L2PF's)
    static std::atomic<unsigned long long> optin{0};        // > 64 KB of dynamic LDS needs an explicit opt-in, per kernel and device
    if (int rc = vlt5_lds_optin(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, WM, WN, AKM, BKM, NS>), (int)lds, optin)) return rc;
#ifdef GEMM_TIMELINE
    const_cast<GemmArgs&>(a).timeline = vlt5_gemm_timeline_buf;
#endif
    if (g_timing.on && g_timing.rec.size() < g_timing.cap) {
        const size_t i = g_timing.rec.size();
        vlt5_gemm_timing_rec r;
        r.M = a.M; r.N = a.N; r.K = a.K; r.batch = (int)grid.z; r.tile_m = BM; r.tile_n = BN; r.a_kmajor = AKM; r.b_kmajor = BKM;
        r.splits = (int)grid.y; r.workgroups = (int)(grid.x * grid.y * grid.z); r.out_f32 = a.out_f32; r.ms = 0.f;
        r.M2 = a.grp_tiles > 0 ? a.gM : 0; r.N2 = a.grp_tiles > 0 ? a.gN : 0;
        r.K2 = a.grp_tiles > 0 ? a.gK : 0; r.batch2 = a.grp_tiles > 0 ? (int)((grid.x - a.grp_tiles) / (a.grp_t2 > 0 ? a.grp_t2 : 1)) : 0;
        if (a.grp_tiles > 0) r.batch = a.grp_tiles / (a.grp_t1 > 0 ? a.grp_t1 : 1);
        g_timing.rec.push_back(r);
        hipExtLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, AKM, BKM, NS>), grid, dim3(WM * WN * 64), lds, st, g_timing.ev[2 * i],
                              g_timing.ev[2 * i + 1], 0, a);
    } else {
        hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, AKM, BKM, NS>), grid, dim3(WM * WN * 64), lds, st, a);
    }
    LAUNCH_CHECK();
    return VLT5_OK;
}

template <int BM, int BN>
int launch_tile(const GemmArgs& a_in, int akm, int bkm, int splits, int batch, hipStream_t st) {
    GemmArgs a = a_in;
    const int t1 = ((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM);
    dim3 grid(t1, splits > 1 ? splits : 1, batch > 1 ? batch : 1);
    if (a.gA) {       // grouped launch: flat grid, every tile of every batch entry of problem 1, then those of problem 2
        const int t2 = ((a.gN + BN - 1) / BN) * ((a.gM + BM - 1) / BM), b1 = batch > 1 ? batch : 1, b2 = a.grp_t2 > 1 ? a.grp_t2 : 1;
        a.grp_tiles = t1 * b1; a.grp_t1 = t1; a.grp_t2 = t2;
        grid = dim3(t1 * b1 + t2 * b2, 1, 1);
    }
    // ring depth: 3 stages up to 64x128 (72 KB, 2 workgroups/CU); 2 for 128x128 (a 3-stage ring = 96 KB = 1 workgroup/CU
    // measured 13 % slower end to end: occupancy matters more) and for 256x256 (2 x 64 KB, one 8-wave workgroup per CU)
#ifndef GEMM_NS_SMALL
#define GEMM_NS_SMALL 3
#endif
    constexpr int NS = (BM + BN <= 192) ? GEMM_NS_SMALL : 2;
    constexpr int WM = 2, WN = (BN == 256) ? 4 : 2;          // 256-wide tiles: the 8-wave kernel
    if (!akm && !bkm) return launch_one<BM, BN, WM, WN, false, false, NS>(a, grid, st);
    if (!akm && bkm) return launch_one<BM, BN, WM, WN, false, true, NS>(a, grid, st);
    if constexpr ((BM * 8) % (WM * WN * 64) == 0) {          // (224 / 160 rows: row-major A only)
        if (akm && bkm) return launch_one<BM, BN, WM, WN, true, true, NS>(a, grid, st);
        return launch_one<BM, BN, WM, WN, true, false, NS>(a, grid, st);
    } else {
        return VLT5_ERR_ARG;
    }
}

#undef g_timing
}  // namespace vlt5gemm
