// Fused encoder self-attention kernels (gfx950): the q|k|v projection of a (sample pair, head pair) and the attention core
// of its four (sample, head) problems in ONE workgroup, q/k/v never leaving the chip between the two.
//
// Reference op sequence being replaced: VL-T5/src/modeling_t5_our.py:282-293 -> HF T5LayerSelfAttention / T5Attention.forward
// (q, k, v = Linear(LN(x)); scores = q k^T (no 1/sqrt(d)) + position_bias + mask; softmax in f32; dropout; . v).
//
// Why this split: an encoder sample has S = L + V <= 56 tokens, a head 64 columns.  A workgroup needs
// M >= 128 rows to make a CU's MFMA rate (4 kFLOP/clk) from the 64 B/clk it can pull into LDS, so it takes TWO samples
// (2 x 64 padded rows); 256 CUs want ~240 workgroups, so it takes TWO heads: 384 weight rows (q|k|v x 2 heads x 64) x d_model.
//   grid  = ceil(B/2) x H/2 workgroups (B = 80, H = 12: 240), 512 threads = 8 waves as 2 (samples) x 4 (96 columns each)
//   phase 1  [128 x d] . [384 x d]^T on MFMA 16x16x32 bf16, BK = 64, two LDS stages filled by global_load_lds (the same
//            interleaved k-step as the 8-wave GEMM: gemm_kernel.h), accumulators 4 x 6 fragments per wave
//   phase 2  accumulators -> bf16 natural tiles Q/K/V[sample][head] in LDS (over the retired stages); q|k|v rows go out to HBM
//            once, coalesced, for the backward (attention backward + weight gradient)
//   phase 3  attention core out of LDS, two waves per (sample, head): scores in registers, softmax over 4-lane groups,
//            dropout, P.V with V^T gathered by ds_read_b64_tr_b16 (attn_core.h) -> ctx rows, row log-sum-exp
// The output projection stays a GEMM of its own: fusing it would split its K = H*64 reduction over the H/2 head pairs and
// trade one 16 us launch for 6 f32 slabs of the residual stream (82 MB written + read per sublayer).
//
// Results are bit-identical to vlt5_gemm_bf16 + vlt5_attn_fwd on the same inputs (same k order, same MFMA, same rounding
// points, same dropout counters): tests/test_gpu_kernels.py::test_fused_qkv_attention_equals_the_unfused_kernels.
#include "gemm_kernel.h"
#include "attn_core.h"
#include <string.h>

namespace {

using vlt5gemm::BK;
using vlt5gemm::lds_off;
using vlt5gemm::lds_ptr_t;
using vlt5attn::AttnArgs;
using vlt5attn::TILE_BYTES;
using vlt5attn::TS;

struct QkvAttnArgs {
    const bf16_t* xn;          // [B*S, d] bf16: LN(x)
    const bf16_t* wqkv;        // [3*inner, d] bf16: q rows | k rows | v rows, head h at rows h*64 .. h*64+63 of each part
    bf16_t* qkv;               // [B*S, 3*inner] bf16 out
    AttnArgs at;               // core: q/k/v strides describe `qkv`; ctx, lse, bias, key_mask, dropout
    int B, S, H, d;
    // T5 RMS norm folded into the kernel (vlt5_qkv_attn_fwd_norm): xn holds bf16(x * w_norm); the q|k|v rows are scaled by
    // rstd[m] = rsqrt(sum of the rs_n partial sums of squares of row m / d + eps) on their way out of the accumulators
    const float* rs_part; int rs_n; float rs_eps; float* rstd_out;
};

// NH = heads per workgroup.  NH = 2: 8 waves as 2 (samples) x 4, a 128 x 384 projection tile, 128 KB of LDS (one workgroup per CU).
// NH = 1: 4 waves as 2 x 2, a 128 x 192 tile, 80 KB of LDS: TWO workgroups share a CU, so the softmax / dropout / store phases of
// one (VALU and memory work with the matrix pipe idle) run under the MFMA main loop of the other.
template <int NH>
struct FusedGeo {
    static constexpr int FBM = 128, FBN = 192 * NH, FWM = 2, FWN = 2 * NH, FNT = 256 * NH;
    static constexpr int FTM = FBM / FWM, FTN = FBN / FWN;          // 64 x 96 per wave
    static constexpr int FFM = FTM / 16, FFN_ = FTN / 16;           // 4 x 6 fragments
    static constexpr int A_BYTES = FBM * BK * 2, B_BYTES = FBN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;   // 16 + 48 (24) KB
    static constexpr int PA = FBM * 8 / FNT, PB = FBN * 8 / FNT, LPT = PA + PB;                             // DMA pieces per wave per k-tile
    static constexpr int FUSED_LDS = 2 * STAGE_BYTES;
    // NH = 2 only (the CU's LDS is the workgroup's own): the score addends -- the relative-position bias blocks of the two heads and the
    // key-mask rows of the two samples -- are staged behind the two stages by LDS-DMA in the prologue (64 gathers of one word per lane
    // after the main loop cost 2.9 k cycles of load issue: profiles/r03_n_enc_attn_timeline.txt)
    static constexpr int XTRA_BIAS = NH == 2 ? 16384 : 0, XTRA = NH == 2 ? XTRA_BIAS + 512 : 0;
    static_assert(6 * NH * TILE_BYTES <= FUSED_LDS, "the Q/K/V tiles of 2 samples x NH heads overlay the retired stages");
};

// TL: debug build of the same kernel that records the shader clock of wave 0 at the phase boundaries (vlt5dbg_qkv_attn_timeline)
template <bool TL, int NH>
__device__ __forceinline__ void qkv_attn_fwd_body(QkvAttnArgs p, unsigned long long* tl_out) {
    using G = FusedGeo<NH>;
    constexpr int FBM = G::FBM, FBN = G::FBN, FWN = G::FWN, FNT = G::FNT, FTM = G::FTM, FTN = G::FTN, FFM = G::FFM, FFN_ = G::FFN_;
    constexpr int A_BYTES = G::A_BYTES, STAGE_BYTES = G::STAGE_BYTES, PA = G::PA, PB = G::PB, LPT = G::LPT;
    (void)FBM; (void)FBN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long tl[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int i) __attribute__((always_inline)) { if (TL && threadIdx.x == 0) tl[i] = __builtin_readcyclecounter(); };
    stamp(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / FWN, wn = wave % FWN;
    const int lrow = lane & 15, lg = lane >> 4;
    const int HP = p.H / NH, inner = p.H * 64;
    // XCD-aware order: workgroup b runs on XCD b % 8; every XCD gets one contiguous run of (sample pair, head pair) tiles, head
    // pair fastest, so the A rows of a sample pair are fetched into ONE L2 and the 6 weight slices (3.5 MB) stay resident in each
    const int ntiles = gridDim.x;
    int tile_id;
    {
        const int b = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, xcd = b & 7, loc = b >> 3;
        tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int b0 = (tile_id / HP) * 2, h0 = (tile_id % HP) * NH;
    const int pi = wave >> 1, cs = NH == 2 ? pi >> 1 : pi, chh = NH == 2 ? pi & 1 : 0;          // phase 3 roles: two waves per (sample, head)
    const int cb = min(b0 + cs, p.B - 1), ch = h0 + chh;

    // per-lane source of each DMA piece at k = 0 (the swizzle of the LDS image is applied to the SOURCE chunk: gemm_kernel.h)
    const bf16_t* src[LPT];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int c = tid + i * FNT, row = c >> 3, kc = (c & 7) ^ (row & 7);
        const int bb = min(b0 + (row >> 6), p.B - 1), t = min(row & 63, p.S - 1);        // padding rows repeat a valid row
        src[i] = p.xn + ((size_t)bb * p.S + t) * p.d + kc * 8;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int c = tid + i * FNT, row = c >> 3, kc = (c & 7) ^ (row & 7);
        const int grow = (row / (64 * NH)) * inner + h0 * 64 + (row % (64 * NH));
        src[PA + i] = p.wqkv + (size_t)grow * p.d + kc * 8;
    }
    // folded norm: a wave's 64 rows are spread over its lanes (row wm*64 + lane); the partial sums of squares are requested first
    // (the oldest loads: they have landed long before the first k-tile) and reduced to rstd behind the prologue's DMA requests
    vlt5gemm::NormRaw rsraw;
    const int rsrow = min(b0 + wm, p.B - 1) * p.S + min(lane, p.S - 1);
    if (p.rs_part) vlt5gemm::norm_request(p.rs_part, p.rs_n, rsrow, rsraw);
    const int wave_base = tid & ~63;
    auto piece = [&](int kt, int s, int pc) __attribute__((always_inline)) {
        char* dst = smem + s * STAGE_BYTES + (pc < PA ? 0 : A_BYTES) + ((pc < PA ? pc : pc - PA) * FNT + wave_base) * 16;
        __builtin_amdgcn_global_load_lds(src[pc] + (size_t)kt * BK, (lds_ptr_t)dst, 16, 0, 0);
    };
    auto ldA = [&](const char* at, int i, int ks) __attribute__((always_inline)) -> bf16x8_t {
        return *reinterpret_cast<const bf16x8_t*>(at + lds_off(wm * FTM + i * 16 + lrow, ks * 4 + lg));
    };
    auto ldB = [&](const char* bt, int j, int ks) __attribute__((always_inline)) -> bf16x8_t {
        return *reinterpret_cast<const bf16x8_t*>(bt + lds_off(wn * FTN + j * 16 + lrow, ks * 4 + lg));
    };

    f32x4_t acc[FFM][FFN_];
#pragma unroll
    for (int i = 0; i < FFM; ++i)
#pragma unroll
        for (int j = 0; j < FFN_; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- phase 1: [128 x d] . [384 x d]^T, BK = 64, two LDS stages, software-pipelined across the k-steps -------------------
    // A k-step is two halves (ks = 0 / 1: the two 32-deep MFMA slabs of the 64-deep tile) with the workgroup barrier BETWEEN them:
    //   ks = 0 half of tile t :  MFMAs on fragments fetched during the previous step; fetch the ks = 1 fragments of tile t
    //   wait (tile t+1 landed; the ks = 1 fragments returned) + barrier  -> stage of tile t is free, stage of tile t+1 is complete
    //   ks = 1 half of tile t :  MFMAs; request tile t+2 into the freed stage (LDS-DMA); fetch the ks = 0 fragments of tile t+1
    // so the matrix pipe never starts a half behind a fragment read (the barrier-at-the-top form of the 8-wave GEMM idles ~450 of
    // its ~2260 cycles per step there), and a DMA piece has a full step to land.  No extra registers: each half fetches into the
    // fragment set the other half has just retired.
    const int nk = p.d / BK;
    auto ks_half = [&](bf16x8_t (&fa)[FFM], bf16x8_t (&fb)[FFN_], bf16x8_t (&na)[FFM], bf16x8_t (&nb)[FFN_],
                       const char* nat, const char* nbt, int nks, int kt_dma, int s_dma, bool dma) __attribute__((always_inline)) {
        // MFMAs of the half held in (fa, fb); behind fragment row i: the reads of the next half's row i (+ its share of the B fragments)
        // and, in a DMA half, two pieces of tile kt_dma.  Placement is left to the compiler from here: hard-pinning (all reads
        // first + one piece behind every third MFMA, or different piece slots for the two waves of a SIMD) measured 9 - 17 % slower
        // -- a piece blocks its wave ~100+ cycles wherever it stands (64 B/clk LDS-fill path, all eight waves in the same phase).
#pragma unroll
        for (int i = 0; i < FFM; ++i) {
#pragma unroll
            for (int j = 0; j < FFN_; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            na[i] = ldA(nat, i, nks);
#pragma unroll
            for (int j = 0; j < FFN_; ++j)
                if (j >= vlt5gemm::bfrag_lo<FFM, FFN_>(i) && j < vlt5gemm::bfrag_lo<FFM, FFN_>(i + 1)) nb[j] = ldB(nbt, j, nks);
            if (dma) {
#pragma unroll
                for (int pc = 0; pc < LPT; ++pc)
                    if (vlt5gemm::piece_row<FFM, LPT, true>(pc) == i) piece(kt_dma, s_dma, pc);
            }
        }
    };
    // the fused [3*inner, d] weight operand was last read a pass ago: every workgroup requests a distinct share of it (one word per
    // 128-byte line) before its own first tile, so the whole panel is on its way into the memory-side cache at once (gemm_kernel.h,
    // GEMM_TOUCH_B); the destination register stays allocated through the main loop, whose waits cover the load
    unsigned touch_sink = 0;
    {
        const int lpr = p.d >> 6;
        const long long nlines = 3LL * inner * lpr, i = (long long)blockIdx.x * FNT + tid;
        // (unconditional -- lanes without a line of their own re-request the first: no control-flow merge, no copy of the register the
        //  load has not landed in yet; gemm_kernel.h)
        const bf16_t* a = i < nlines ? p.wqkv + (size_t)(i / lpr) * p.d + (size_t)(i % lpr) * 64 : p.wqkv;
        asm volatile("global_load_dword %0, %1, off" : "=v"(touch_sink) : "v"(a));
    }
#pragma unroll
    for (int pc = 0; pc < LPT; ++pc) piece(0, 0, pc);
#pragma unroll
    for (int pc = 0; pc < LPT; ++pc) piece(min(1, nk - 1), 1, pc);
    // score addends into LDS (behind the stages): 64 words per piece, source clamped to the operand's last word; the first main-loop
    // wait (vmcnt(0) + barrier) covers them for every wave
    const int nbias = p.at.bias ? NH * p.at.bias_q * p.at.bias_k : 0;
    const bool add_lds = G::XTRA > 0 && nbias * 4 <= G::XTRA_BIAS && p.at.Tk <= 64;
    float* const bias_l = reinterpret_cast<float*>(smem + G::FUSED_LDS);
    float* const mask_l = reinterpret_cast<float*>(smem + G::FUSED_LDS + G::XTRA_BIAS);                 // [2 samples][64]
    if (add_lds) {
        const float* bsrc = p.at.bias ? p.at.bias + (size_t)h0 * p.at.bias_q * p.at.bias_k : nullptr;
        for (int c = wave; c * 64 < nbias; c += FNT / 64)
            __builtin_amdgcn_global_load_lds(bsrc + min(c * 64 + lane, nbias - 1), (lds_ptr_t)(reinterpret_cast<char*>(bias_l) + c * 256), 4, 0, 0);
        if (p.at.key_mask && wave < 2)
            __builtin_amdgcn_global_load_lds(p.at.key_mask + (size_t)min(b0 + wave, p.B - 1) * p.at.Tk + min(lane, p.at.Tk - 1),
                                             (lds_ptr_t)(reinterpret_cast<char*>(mask_l) + wave * 256), 4, 0, 0);
    }
    stamp(1);
    float rs_own = 1.f;
    if (p.rs_part) {
        rs_own = vlt5gemm::norm_reduce(rsraw, p.rs_n, 1.0f / (float)p.d, p.rs_eps);
        if (p.rstd_out && h0 == 0 && wn == 0 && b0 + wm < p.B && lane < p.S) p.rstd_out[rsrow] = rs_own;
    }
    bf16x8_t fa0[FFM], fb0[FFN_], fa1[FFM], fb1[FFN_];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");          // tile 0 (the older group) has landed
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < FFM; ++i) fa0[i] = ldA(smem, i, 0);
#pragma unroll
    for (int j = 0; j < FFN_; ++j) fb0[j] = ldB(smem + A_BYTES, j, 0);
    int stage = 0;
    for (int it = 0; it < nk; ++it) {
        const char* at = smem + stage * STAGE_BYTES;
        const char* bt = at + A_BYTES;
        const char* at2 = smem + (stage ^ 1) * STAGE_BYTES;
        const char* bt2 = at2 + A_BYTES;
        ks_half(fa0, fb0, fa1, fb1, at, bt, 1, 0, 0, false);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (TL && it == 0) stamp(2);
        ks_half(fa1, fb1, fa0, fb0, at2, bt2, 0, min(it + 2, nk - 1), stage, true);   // (past the end: re-request the last tile)
        stage ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the surplus prefetch must land before the stages are reused
    asm volatile("" ::"v"(touch_sink));
    stamp(3);
    // score addends (relative-position bias + key mask) of this wave's two 16-row blocks: requested here (branch-free, all loads in
    // flight together) and combined after the hand-over, which hides their global round trip; kept out of the main loop, whose
    // register budget is full
    vlt5attn::AddendRaw raw[2];
    if (!add_lds) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) vlt5attn::score_addend_load(p.at, cb, ch, (wave & 1) * 32 + blk * 16, lane, raw[blk]);
    } else {
        // out of the LDS copies (same words, same arithmetic afterwards): the bias block of head ch and the mask row of sample cs.  Read
        // BEFORE the hand-over barrier: the accumulator tiles overlay the stages only, the copies live behind them
        const float* brow = bias_l + (size_t)chh * p.at.bias_q * p.at.bias_k;
        const float* mrow = mask_l + cs * 64;
        const bool hb = p.at.bias != nullptr, hm = p.at.key_mask != nullptr;
        const int bk1 = hb ? p.at.bias_k - 1 : 0, mk1 = hm ? p.at.Tk - 1 : 0;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int i = (wave & 1) * 32 + blk * 16 + lrow;
            const float* row = brow + (hb ? min(i, p.at.bias_q - 1) * p.at.bias_k : 0);
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    raw[blk].bv[jb][r] = row[min(jb * 16 + lg * 4 + r, bk1)];
                    raw[blk].mv[jb][r] = mrow[min(jb * 16 + lg * 4 + r, mk1)];
                }
        }
    }
    stamp(8);
    __syncthreads();
    stamp(9);

    // ---- phase 2: accumulators -> bf16 natural tiles; tile (s, hh, part) at ((s*2 + hh)*3 + part) * TILE_BYTES ---------------
    float rsc[FFM];                                          // rstd of fragment row i*16 + lrow: held by lane i*16 + lrow of this wave
#pragma unroll
    for (int i = 0; i < FFM; ++i) rsc[i] = p.rs_part ? __shfl(rs_own, i * 16 + lrow, 64) : 1.f;
    auto tile = [&](int s, int hh, int part) __attribute__((always_inline)) -> bf16_t* {
        return reinterpret_cast<bf16_t*>(smem + ((s * NH + hh) * 3 + part) * TILE_BYTES);
    };
#pragma unroll
    for (int j = 0; j < FFN_; ++j) {
        const int n = wn * FTN + j * 16;                       // 16 | 64: a fragment lies inside one (part, head) tile
        bf16_t* tl = tile(wm, NH == 2 ? (n >> 6) & 1 : 0, n / (64 * NH)) + (n & 63) + lg * 4;
#pragma unroll
        for (int i = 0; i < FFM; ++i) {
            uint2 pk;
            pk.x = pack_bf16x2(acc[i][j][0] * rsc[i], acc[i][j][1] * rsc[i]);
            pk.y = pack_bf16x2(acc[i][j][2] * rsc[i], acc[i][j][3] * rsc[i]);
            *reinterpret_cast<uint2*>(tl + (i * 16 + lrow) * TS) = pk;
        }
    }
    stamp(10);
    __syncthreads();
    stamp(4);
    // q | k | v rows to HBM (saved for the backward): 6 * NH tiles x S rows x 128 bytes; 8 lanes per row, FNT / 8 rows per pass
    {
        constexpr int RPP = FNT / 8;                            // rows per pass: 64 (8 waves) / 32 (4 waves)
        const int ch8 = (tid & 7) * 8;
#pragma unroll
        for (int t0 = 0; t0 < 64; t0 += RPP) {
            const int t = t0 + (tid >> 3);
            if (t < p.S) {
#pragma unroll
                for (int ti = 0; ti < 6 * NH; ++ti) {
                    const int s = ti / (3 * NH), hh = (ti / 3) % NH, part = ti % 3;
                    if (b0 + s < p.B) {
                        const uint4 v = *reinterpret_cast<const uint4*>(smem + ti * TILE_BYTES + (t * TS + ch8) * 2);
                        *reinterpret_cast<uint4*>(p.qkv + ((size_t)(b0 + s) * p.S + t) * (3 * inner) + part * inner + (h0 + hh) * 64 + ch8) = v;
                    }
                }
            }
        }
    }
    stamp(5);
    // ---- phase 3: attention core, two waves per (sample, head), 32 query rows each: the two 16-row blocks are independent
    // chains in one basic block (rows beyond S compute on padding and store nothing) -------------------------------------------
    if (b0 + cs < p.B) {
        float add[2][4][4];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) vlt5attn::score_addend_finish(p.at, (wave & 1) * 32 + blk * 16, lane, raw[blk], add[blk]);
        auto cstamp = [&](int i) __attribute__((always_inline)) {
            if (TL && threadIdx.x == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tl[12 + i] = __builtin_readcyclecounter(); }
        };
        stamp(11);
        vlt5attn::attn_fwd_blocks<true, 2>(p.at, tile(cs, chh, 0), tile(cs, chh, 1), tile(cs, chh, 2), cb, ch, (wave & 1) * 32, lane, add, cstamp);
    }
    stamp(6);
    if (TL) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(7);
        if (threadIdx.x == 0 && tl_out)
            for (int q = 0; q < 16; ++q) tl_out[(size_t)blockIdx.x * 16 + q] = tl[q];
    }
}

// one head per workgroup: 4 waves, registers capped for two waves per SIMD (two workgroups per CU); two heads: 8 waves
template <bool TL>
__global__ __launch_bounds__(256, 2) void qkv_attn_fwd_kernel(QkvAttnArgs p, unsigned long long* tl_out) { qkv_attn_fwd_body<TL, 1>(p, tl_out); }
template <bool TL>
__global__ __launch_bounds__(512) void qkv_attn_fwd2_kernel(QkvAttnArgs p, unsigned long long* tl_out) { qkv_attn_fwd_body<TL, 2>(p, tl_out); }

unsigned long long* g_tl_buf = nullptr;      // vlt5dbg_qkv_attn_timeline: device buffer of 8 x u64 per workgroup, or null


// q|k|v projection + attention core of one encoder self-attention (the kernel above).  Shapes: d_kv = 64, H even, S <= 64,
// d_model a multiple of 64; the strides in `core` must describe `qkv` ([B, S, 3*H*64], q | k | v).
template <int NH, class KF, class KT>
static int qkv_attn_dispatch(const QkvAttnArgs& a, KF kernel, KT kernel_tl, void* stream) {
    using G = FusedGeo<NH>;
    static std::atomic<unsigned long long> optin_a{0}, optin_b{0};      // devices on which the kernels may use FUSED_LDS bytes of LDS
    {
        int rc = vlt5_lds_optin(reinterpret_cast<const void*>(kernel), G::FUSED_LDS + G::XTRA, optin_a);
        if (rc) return rc;
        rc = vlt5_lds_optin(reinterpret_cast<const void*>(kernel_tl), G::FUSED_LDS + G::XTRA, optin_b);
        if (rc) return rc;
    }
    const int inner = a.H * 64;
    const int grid = ((a.B + 1) / 2) * (a.H / NH);
    vlt5gemm::TimingState& tm = vlt5_gemm_timing_state;           // bench.py's in-situ roofline covers this MFMA kernel too
    if (g_tl_buf) hipLaunchKernelGGL(kernel_tl, dim3(grid), dim3(G::FNT), G::FUSED_LDS + G::XTRA, (hipStream_t)stream, a, g_tl_buf);
    else if (tm.on && tm.rec.size() < tm.cap) {
        const size_t i = tm.rec.size();
        vlt5_gemm_timing_rec r;
        r.M = a.B * a.S; r.N = 3 * inner; r.K = a.d; r.batch = 1; r.tile_m = G::FBM; r.tile_n = 384; r.a_kmajor = 0; r.b_kmajor = 0;
        r.splits = 1; r.workgroups = grid; r.out_f32 = 0; r.ms = 0.f; r.M2 = 0; r.N2 = 0; r.K2 = 0; r.batch2 = 0;
        tm.rec.push_back(r);
        hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(G::FNT), G::FUSED_LDS + G::XTRA, (hipStream_t)stream, tm.ev[2 * i], tm.ev[2 * i + 1], 0, a,
                              (unsigned long long*)nullptr);
    } else hipLaunchKernelGGL(kernel, dim3(grid), dim3(G::FNT), G::FUSED_LDS + G::XTRA, (hipStream_t)stream, a, (unsigned long long*)nullptr);
    LAUNCH_CHECK();
    return VLT5_OK;
}
}  // namespace

static int qkv_attn_launch(const void* xn_bf16, const void* wqkv_bf16, void* qkv_bf16, const vlt5_attn_desc* core, int d_model,
                           const float* norm_partials, int norm_nparts, float norm_eps, float* norm_rstd_out, void* stream);
extern "C" int vlt5_qkv_attn_fwd(const void* xn_bf16, const void* wqkv_bf16, void* qkv_bf16, const vlt5_attn_desc* core, int d_model,
                                 void* stream) {
    return qkv_attn_launch(xn_bf16, wqkv_bf16, qkv_bf16, core, d_model, nullptr, 0, 0.f, nullptr, stream);
}
// the same kernel with the T5 RMS norm in front of the projection folded in: xw_bf16 = bf16(x * w_norm) and the per-row partial
// sums of squares of x as the producing GEMM's epilogue left them (vlt5_gemm_desc.emit_*); the q|k|v rows are scaled by rstd on
// their way out of the accumulators and rstd goes to norm_rstd_out [B*S] for the backward.  norm_nparts <= 32 (SSQ_STRIDE).
extern "C" int vlt5_qkv_attn_fwd_norm(const void* xw_bf16, const void* wqkv_bf16, void* qkv_bf16, const vlt5_attn_desc* core, int d_model,
                                      const float* norm_partials, int norm_nparts, float norm_eps, float* norm_rstd_out, void* stream) {
    if (!norm_partials || norm_nparts < 1 || norm_nparts > 32 || (((uintptr_t)norm_partials) & 15)) return VLT5_ERR_ARG;
    return qkv_attn_launch(xw_bf16, wqkv_bf16, qkv_bf16, core, d_model, norm_partials, norm_nparts, norm_eps, norm_rstd_out, stream);
}
static int qkv_attn_launch(const void* xn_bf16, const void* wqkv_bf16, void* qkv_bf16, const vlt5_attn_desc* core, int d_model,
                           const float* norm_partials, int norm_nparts, float norm_eps, float* norm_rstd_out, void* stream) {
    if (!xn_bf16 || !wqkv_bf16 || !qkv_bf16 || !core || !core->ctx) return VLT5_ERR_ARG;
    if (core->dk != 64 || core->Tq != core->Tk || core->Tq < 1 || core->Tq > 64 || core->B < 1 || d_model < 64) return VLT5_ERR_ARG;
    if (d_model & 63) return VLT5_ERR_ALIGN;
    const int inner = core->H * 64, S = core->Tq;
    const bf16_t* base = (const bf16_t*)qkv_bf16;
    if (core->q != (const void*)base || core->k != (const void*)(base + inner) || core->v != (const void*)(base + 2 * inner)) return VLT5_ERR_ARG;
    if (core->q_st != 3 * inner || core->k_st != 3 * inner || core->v_st != 3 * inner) return VLT5_ERR_ARG;
    if (core->q_sb != (long long)S * 3 * inner || core->k_sb != core->q_sb || core->v_sb != core->q_sb) return VLT5_ERR_ARG;
    if ((core->o_sb & 3) || (core->o_st & 3)) return VLT5_ERR_ALIGN;
    QkvAttnArgs a;
    a.xn = (const bf16_t*)xn_bf16; a.wqkv = (const bf16_t*)wqkv_bf16; a.qkv = (bf16_t*)qkv_bf16;
    a.B = core->B; a.S = S; a.H = core->H; a.d = d_model;
    a.rs_part = norm_partials; a.rs_n = norm_nparts; a.rs_eps = norm_eps; a.rstd_out = norm_rstd_out;
    AttnArgs& t = a.at;
    t.q = base; t.k = base + inner; t.v = base + 2 * inner;
    t.q_sb = core->q_sb; t.q_st = core->q_st; t.k_sb = core->k_sb; t.k_st = core->k_st; t.v_sb = core->v_sb; t.v_st = core->v_st;
    t.ctx = (bf16_t*)core->ctx; t.o_sb = core->o_sb; t.o_st = core->o_st; t.lse = core->lse;
    t.bias = core->bias; t.bias_q = core->bias_q; t.bias_k = core->bias_k;
    t.key_mask = core->key_mask; t.mask_value = core->mask_value; t.causal = core->causal;
    t.B = core->B; t.H = core->H; t.Tq = S; t.Tk = S; t.dk = 64;
    t.drop_thr = core->drop_p > 0.f ? drop_thr16(core->drop_p) : 0u; t.drop_seed = core->drop_seed;
    t.d_ctx = nullptr; t.do_sb = t.do_st = 0; t.dq = t.dk_ = t.dv = nullptr;
    t.dq_sb = t.dq_st = t.dk_sb = t.dk_st = t.dv_sb = t.dv_st = 0; t.dbias = nullptr;
    // one head per workgroup (4 waves, two workgroups per CU) or two (8 waves, one per CU): vlt5_attn_desc.fused_heads = 1 / 2 is an
    // experiment switch (0: two); an odd number of heads can only run the one-head kernel
    const int nh = (core->fused_heads != 1 && !(a.H & 1)) ? 2 : 1;
    if (nh == 2) return qkv_attn_dispatch<2>(a, &qkv_attn_fwd2_kernel<false>, &qkv_attn_fwd2_kernel<true>, stream);
    return qkv_attn_dispatch<1>(a, &qkv_attn_fwd_kernel<false>, &qkv_attn_fwd_kernel<true>, stream);
}

// The whole encoder self-attention sublayer forward (SURVEY 8(b) `vlt5_enc_attn_fwd`; HF T5LayerSelfAttention.forward as called from
// VL-T5/src/modeling_t5_our.py:282-293):  x_out = x + dropout( o( softmax(q k^T + bias + mask) v ) ),  q|k|v = LN(x) Wqkv^T.
// Three launches: T5 RMS norm (emits the bf16 operand + rstd), the fused projection + core kernel, the output projection with its
// dropout + residual epilogue.  Everything the backward needs is left in the caller's buffers.
extern "C" int vlt5_enc_attn_fwd(const vlt5_enc_attn_desc* d, void* stream) {
    if (!d || !d->x || !d->ln_w || !d->wqkv_bf16 || !d->wo_bf16 || !d->x_out || !d->xn_bf16 || !d->rstd || !d->qkv_bf16 || !d->ctx_bf16 || !d->lse)
        return VLT5_ERR_ARG;
    const int M = d->B * d->S, inner = d->H * 64;
    int rc = vlt5_layernorm_fwd(d->x, d->ln_w, d->xn_bf16, nullptr, d->rstd, M, d->d_model, d->eps, 0.f, 0, 0, 0, stream);
    if (rc) return rc;
    vlt5_attn_desc a;
    memset(&a, 0, sizeof a);
    bf16_t* qkv = (bf16_t*)d->qkv_bf16;
    a.q = qkv; a.k = qkv + inner; a.v = qkv + 2 * inner;
    a.q_sb = a.k_sb = a.v_sb = (long long)d->S * 3 * inner; a.q_st = a.k_st = a.v_st = 3 * inner;
    a.ctx = d->ctx_bf16; a.o_sb = (long long)d->S * inner; a.o_st = inner; a.lse = d->lse;
    a.bias = d->bias; a.bias_q = d->bias_q; a.bias_k = d->bias_k; a.key_mask = d->key_mask; a.mask_value = d->mask_value;
    a.B = d->B; a.H = d->H; a.Tq = d->S; a.Tk = d->S; a.dk = 64; a.drop_p = d->drop_p; a.drop_seed = d->seed_probs;
    rc = vlt5_qkv_attn_fwd(d->xn_bf16, d->wqkv_bf16, d->qkv_bf16, &a, d->d_model, stream);
    if (rc) return rc;
    vlt5_gemm_desc g;
    memset(&g, 0, sizeof g);
    g.A = d->ctx_bf16; g.B = d->wo_bf16; g.C = d->x_out; g.M = M; g.N = d->d_model; g.K = inner; g.lda = inner; g.ldb = inner; g.ldc = d->d_model;
    g.alpha = 1.f; g.out_f32 = 1; g.resid = d->x; g.ldr = d->d_model; g.drop_p = d->drop_p; g.drop_seed = d->seed_out;
    return vlt5_gemm_bf16(&g, stream);
}

// Backward of the sublayer (SURVEY 8(b) `vlt5_enc_attn_bwd`; autograd's backward of HF T5LayerSelfAttention as reached from
// src/vqacl.py:461), from what vlt5_enc_attn_fwd saved:
//   dyd = bf16(dropout'(dy))                      d ctx = dyd Wo          dWo   = dyd^T ctx
//   d q|k|v (+ dS for the relative-position table) = attention backward     dWqkv = dqkv^T xn
//   dxn = dqkv Wqkv                                dx = dy + T5-RMS-norm backward(dxn),  d ln_w
// Scratch: vlt5_enc_attn_bwd_workspace_bytes(B, S, H, d_model) bytes of device memory owned by the caller.
extern "C" long long vlt5_enc_attn_bwd_workspace_bytes(int B, int S, int H, int d_model) {
    if (B < 1 || S < 1 || H < 1 || d_model < 1) return -1;
    const long long M = (long long)B * S, inner = (long long)H * 64;
    auto up = [](long long x) { return (x + 255) / 256 * 256; };
    return up(M * d_model * 2) + up(M * inner * 2) + up(M * 3 * inner * 2) + up(M * d_model * 4) +
           up((long long)vlt5_layernorm_bwd_blocks((int)M) * d_model * 4);
}
extern "C" int vlt5_enc_attn_bwd(const vlt5_enc_attn_desc* d, const vlt5_enc_attn_grads* g, void* workspace, void* stream) {
    if (!d || !g || !workspace || !d->x || !d->ln_w || !d->wqkv_bf16 || !d->wo_bf16 || !d->xn_bf16 || !d->rstd || !d->qkv_bf16 ||
        !d->ctx_bf16 || !d->lse || !g->dy || !g->dx || !g->d_wqkv || !g->d_wo || !g->d_ln_w)
        return VLT5_ERR_ARG;
    const int M = d->B * d->S, inner = d->H * 64, dm = d->d_model;
    auto up = [](long long x) { return (x + 255) / 256 * 256; };
    char* ws = (char*)workspace;
    bf16_t* dyd = (bf16_t*)ws;              ws += up((long long)M * dm * 2);
    bf16_t* dctx = (bf16_t*)ws;             ws += up((long long)M * inner * 2);
    bf16_t* dqkv = (bf16_t*)ws;             ws += up((long long)M * 3 * inner * 2);
    float* dxn = (float*)ws;                ws += up((long long)M * dm * 4);
    float* lnpart = (float*)ws;
    int rc = vlt5_drop_cast(g->dy, dyd, M, dm, d->drop_p, d->seed_out, stream);
    if (rc) return rc;
    vlt5_gemm_desc m;
    auto gemm = [&](const void* A, const void* Bm, void* Cc, int Mm, int Nn, int Kk, int lda, int ldb, int ldc, int akm, int bkm, int f32) {
        memset(&m, 0, sizeof m);
        m.A = A; m.B = Bm; m.C = Cc; m.M = Mm; m.N = Nn; m.K = Kk; m.lda = lda; m.ldb = ldb; m.ldc = ldc;
        m.a_kmajor = akm; m.b_kmajor = bkm; m.alpha = 1.f; m.out_f32 = f32;
        return vlt5_gemm_bf16(&m, stream);
    };
    // d ctx = dyd Wo (Wo [d_model, inner] read k-major);  dWo [d_model, inner] = dyd^T ctx
    if ((rc = gemm(dyd, d->wo_bf16, dctx, M, inner, dm, dm, inner, inner, 0, 1, 0))) return rc;
    if ((rc = gemm(dyd, d->ctx_bf16, g->d_wo, dm, inner, M, dm, inner, inner, 1, 1, 1))) return rc;
    vlt5_attn_desc a;
    memset(&a, 0, sizeof a);
    bf16_t* qkv = (bf16_t*)d->qkv_bf16;
    a.q = qkv; a.k = qkv + inner; a.v = qkv + 2 * inner;
    a.q_sb = a.k_sb = a.v_sb = (long long)d->S * 3 * inner; a.q_st = a.k_st = a.v_st = 3 * inner;
    a.lse = d->lse; a.bias = d->bias; a.bias_q = d->bias_q; a.bias_k = d->bias_k; a.key_mask = d->key_mask; a.mask_value = d->mask_value;
    a.B = d->B; a.H = d->H; a.Tq = d->S; a.Tk = d->S; a.dk = 64; a.drop_p = d->drop_p; a.drop_seed = d->seed_probs;
    a.d_ctx = dctx; a.do_sb = (long long)d->S * inner; a.do_st = inner;
    a.dq = dqkv; a.dk_ = dqkv + inner; a.dv = dqkv + 2 * inner;
    a.dq_sb = a.dk_sb = a.dv_sb = (long long)d->S * 3 * inner; a.dq_st = a.dk_st = a.dv_st = 3 * inner;
    a.dbias = g->d_scores;
    if ((rc = vlt5_attn_bwd(&a, stream))) return rc;
    // dWqkv [3*inner, d_model] = dqkv^T xn;  dxn = dqkv Wqkv (Wqkv [3*inner, d_model] read k-major)
    if ((rc = gemm(dqkv, d->xn_bf16, g->d_wqkv, 3 * inner, dm, M, 3 * inner, dm, dm, 1, 1, 1))) return rc;
    if ((rc = gemm(dqkv, d->wqkv_bf16, dxn, M, dm, 3 * inner, 3 * inner, dm, dm, 0, 1, 1))) return rc;
    if (g->dx != g->dy) HIP_RET(hipMemcpyAsync(g->dx, g->dy, (size_t)M * dm * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return vlt5_layernorm_bwd(dxn, d->x, d->ln_w, d->rstd, g->dx, g->d_ln_w, lnpart, M, dm, 1, 0, 0.f, 0, 0, 0, nullptr, 0.f, 0, stream);
}

// debug (not part of the public ABI): while `buf` (device, 8 x u64 per workgroup) is set, vlt5_qkv_attn_fwd launches the
// instrumented build of its kernel: [0] start, [1] prologue issued, [2] first k-tile landed, [3] main loop done, [4] tiles
// handed over, [5] q|k|v stores issued, [6] core done, [7] stores drained (shader clocks of wave 0); tools/enc_attn_timeline.py
extern "C" int vlt5dbg_qkv_attn_timeline(void* buf) { g_tl_buf = (unsigned long long*)buf; return VLT5_OK; }
