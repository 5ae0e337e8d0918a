// Fused decoder attention sublayers (gfx950): projection + attention core + per-head share of the output projection in ONE
// workgroup per (sample group, head) -- SURVEY 8(b) `vlt5_dec_self_attn_fwd` / `vlt5_cross_attn_fwd`.
//
// Reference op sequence being replaced: HF T5LayerSelfAttention / T5LayerCrossAttention.forward as reached from
// VL-T5/src/modeling_t5_our.py:641-655 (decoder T5Stack): q (k, v) = Linear(LN(x)); scores = q k^T + position_bias (+ causal
// mask) / + encoder key mask; softmax in f32; dropout; . v; o = Linear(ctx); x + dropout(o).
//
// Why: the decoder is B*T = 400 rows.  Every launch on it is 5 - 8 us of latency (launch boundary, kernel arguments, first bytes
// from L2, stores landing) around ~1 us of work, and a sublayer was three of them (projection GEMM, core, output projection).
// Here a workgroup owns 32 rows (floor(32/T) samples) of ONE head for the whole chain, so nothing between the three steps
// crosses a launch:
//   grid     ceil(B / SG) * H workgroups (B = 80, T = 5: 14 * 12 = 168), 512 threads = 8 waves
//   phase 1  [32 x d] . [192 (q|k|v of head h) x d]^T  (cross: [64 (q) x d]^T) on MFMA 16x16x32 bf16, BK = 64, LDS ring filled by
//            global_load_lds; the k order, MFMA and rounding points are those of vlt5_gemm_bf16 -> q|k|v bit-identical to it
//   hand-over accumulators -> bf16 natural tiles in LDS; the rows also go to HBM for the backward
//   phase 2  attention core (attn_core.h, the stand-alone kernel's code), one wave per sample, ctx rows to HBM and into an LDS tile;
//            cross-attention: K fragments straight from global memory (requested at kernel start), V tiles staged in LDS
//   phase 3  [32 x 64] ctx . [d x 64]^T of Wo (columns of head h), prefetched into the retired ring during the core -> f32 slab
//            `h` of the output projection.  The H slabs are summed in fixed order, with dropout and the residual, by the norm
//            kernel that follows (vlt5_layernorm_fwd_slabs) -- as for the split-K output projections this replaces.
// The core's results are bit-identical to vlt5_gemm_bf16 + vlt5_attn_fwd; the output projection differs from the GEMM's by the
// f32 summation order over heads only (tests/test_gpu_kernels.py::test_fused_decoder_attention_sublayers).
#include "gemm_kernel.h"
#include "attn_core.h"
#include <string.h>
#include <algorithm>

namespace {

using vlt5attn::AttnArgs;
using vlt5attn::TS;
using vlt5gemm::BK;
using vlt5gemm::lds_off;
using vlt5gemm::lds_ptr_t;

struct DecAttnArgs {
    const bf16_t* xn;          // [B*T, d] bf16: LN(x)
    const bf16_t* w;           // self: [3*inner, d] q | k | v rows; cross: [inner, d] q rows
    const bf16_t* wo;          // [d, inner]
    bf16_t* qkv;               // self: [B*T, 3*inner] out; cross: [B*T, inner] out (q)
    float* slabs;              // [H][slab_stride] f32 out: slab h = ctx_h . Wo[:, h*64 .. h*64+63]^T, rows [B*T, d]
    long long slab_stride;
    AttnArgs at;               // core; cross: k / v and their strides describe the encoder-side projections
    int B, T, H, d, SG;        // SG = samples per workgroup (SG * T <= 32)
};

constexpr int DNT = 512;
constexpr int ROWS = 32;
constexpr int A_BYTES = ROWS * BK * 2;               // 4 KB
constexpr int CHUNK_ROWS = 192;                      // rows of Wo per phase-3 chunk (two waves x 96)
constexpr int CHUNK_BYTES = CHUNK_ROWS * BK * 2;     // 24 KB
constexpr int NCHUNK = 4;                            // chunks resident at once
constexpr int QROWS = 48;                            // Q tile rows: the last sample's 16-row block may reach row 32 - T + 15
constexpr int CTX_BYTES = ROWS * BK * 2;

template <bool CROSS>
struct DecGeo {
    static constexpr int BROWS = CROSS ? 64 : 192;
    static constexpr int B_BYTES = BROWS * BK * 2;
    static constexpr int STAGE = A_BYTES + B_BYTES;                    // 12 / 28 KB
    static constexpr int SLOTS = STAGE / 16;                           // 768 / 1792 sixteen-byte slots
    static constexpr int P = (SLOTS + DNT - 1) / DNT;                  // 2 / 4 DMA pieces per k-tile; the last one: threads < 256 only
    static constexpr int NST = CROSS ? 8 : 5;                          // k-tiles in flight: NST - 1 (84 / 112 KB per workgroup)
    static constexpr int REGION = NST * STAGE;                         // ring; later the Wo chunks (self: + the Q / K / V tiles): 96 / 140 KB
    static_assert(NCHUNK * CHUNK_BYTES + (CROSS ? 0 : (QROWS + 2 * 96) * vlt5attn::TS * 2) <= REGION, "what follows the ring fits into it");
    static_assert(SLOTS % DNT == DNT / 2, "last piece is half a workgroup wide");
    static_assert(NST - 2 <= 6, "the k-loop spells out its counted waits up to 6 young tiles");
};
constexpr int SELF_KV_ROWS = 96;                      // K / V tile rows of the self-attention variant (sample offset + 63 < 96)
constexpr int SELF_LDS = DecGeo<false>::REGION + CTX_BYTES;
constexpr int MAXSPB = 3;                             // samples whose query rows share one 16-row block of the core
constexpr int CROSS_FIXED_LDS = DecGeo<true>::REGION + QROWS * TS * 2 + CTX_BYTES;
constexpr int LDS_MAX = 160 * 1024;

struct KFromRegs {
    bf16x8_t f[4][2];
    __device__ __forceinline__ bf16x8_t operator()(int jb, int ks) const { return f[jb][ks]; }
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;      // 16 bytes in four VGPRs (usable as an asm operand)
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- attention core of ONE 16-row block that holds the query rows of up to MAXSPB samples (T <= 5: three) -------------------------
// A sample of the decoder has T <= 16 query rows, a wave's MFMA block 16: run per sample (attn_core.h attn_fwd_blocks), two thirds of
// the lanes of every softmax / dropout instruction work on padding, and six waves share four SIMDs.  Here the rows of several
// samples share the block: the scores are taken per sample (its own K; a lane keeps the result of ITS sample), softmax, dropout and
// packing run once, P.V accumulates per sample with the probabilities of the other samples' rows zeroed (exact zeros into the
// accumulator).  Per row the arithmetic and its order are those of attn_fwd_blocks -- results bit-identical to the stand-alone kernel.
// NJB = 16-key blocks that can hold keys (1: Tk <= 16, the causal self-attention; 4: Tk <= 64).
template <int NJB, class KF>
__device__ __forceinline__ void dec_core_block(const AttnArgs& p, int h, int lane, int nsb, int T, int bfirst, const bf16_t* Qb,
                                               const KF (&kf)[MAXSPB], const bf16_t* const (&Vs)[MAXSPB], const float (&add)[4][4],
                                               char* ctxt, int ctx_row0) {
    const int lr = lane & 15, g = lane >> 4;
    const bool valid = lr < nsb * T;
    const int sl = valid ? lr / T : 0, i = valid ? lr - sl * T : 0, b = bfirst + sl;       // this lane's row: sample slot, query index
    bf16x8_t fq[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fq[ks] = vlt5attn::lds_frag(Qb, lr, ks * 4 + g);
    f32x4_t acc[NJB];
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) acc[jb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s3 = 0; s3 < MAXSPB; ++s3) {
        if (s3 < nsb) {
            f32x4_t a[NJB];
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) a[jb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int jb = 0; jb < NJB; ++jb) a[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[s3](jb, ks), fq[ks], a[jb], 0, 0, 0);
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[jb][r] = sl == s3 ? a[jb][r] : acc[jb][r];
        }
    }
    const float dsc = drop_scale(p.drop_thr);
    constexpr float kLog2e = 1.4426950408889634f;
    const uint32_t rowbase = ((uint32_t)(b * p.H + h) * (uint32_t)p.Tq + (uint32_t)i) * (uint32_t)p.Tk + (uint32_t)(g * 4);
    float s[4][4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[jb][r] = jb < NJB ? acc[jb < NJB ? jb : 0][r] + add[jb][r] : 0.f;
    float mj[4];
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) mj[jb] = fmaxf(fmaxf(s[jb][0], s[jb][1]), fmaxf(s[jb][2], s[jb][3]));
    const float m = vlt5attn::quad_lane_max(NJB == 4 ? fmaxf(fmaxf(mj[0], mj[1]), fmaxf(mj[2], mj[3])) : mj[0]);
    const float m2 = m * kLog2e;
    float sj[4];
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[jb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[jb][r], kLog2e, -m2));
        sj[jb] = (s[jb][0] + s[jb][1]) + (s[jb][2] + s[jb][3]);
    }
    // (key blocks past NJB hold exp(-inf) = 0 in the stand-alone kernel: x + 0 is x, so the shorter sum is the same number)
    const float sum = vlt5attn::quad_lane_sum(NJB == 4 ? (sj[0] + sj[1]) + (sj[2] + sj[3]) : sj[0]);
    const float vs = __builtin_amdgcn_rcpf(sum) * dsc;
    if (g == 0 && valid && p.lse) p.lse[((size_t)b * p.H + h) * p.Tq + i] = m + logf(sum);
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) {
        bool keep[4] = {true, true, true, true};
        if (p.drop_thr) drop_keep4(p.drop_seed, rowbase + jb * 16, p.drop_thr, keep);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[jb][r] = keep[r] ? s[jb][r] * vs : 0.f;
    }
    bf16x8_t pf[2];
    pf[0] = vlt5attn::pack_slots(s[0], s[1]);
    pf[1] = vlt5attn::pack_slots(s[2], s[3]);
    const bf16x8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        f32x4_t o = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s3 = 0; s3 < MAXSPB; ++s3) {
            if (s3 < nsb) {
#pragma unroll
                for (int kk = 0; kk < (NJB == 4 ? 2 : 1); ++kk) {
                    if (kk * 32 < p.Tk) {
                        const bf16x8_t fv = vlt5attn::frag_tr_slots(Vs[s3], db * 16, kk, lane);
                        o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, sl == s3 ? pf[kk] : zero8, o, 0, 0, 0);
                    }
                }
            }
        }
        const int d = db * 16 + g * 4;
        if (valid) {
            uint2 pk;
            pk.x = pack_bf16x2(o[0], o[1]);
            pk.y = pack_bf16x2(o[2], o[3]);
            *reinterpret_cast<uint2*>(p.ctx + b * p.o_sb + (long long)i * p.o_st + (long long)h * 64 + d) = pk;
            *reinterpret_cast<uint2*>(ctxt + lds_off(ctx_row0 + lr, d >> 3) + (d & 7) * 2) = pk;
        }
    }
}

// TL: debug build of the same kernel that records the shader clock of wave 0 at the phase boundaries (vlt5dbg_dec_attn_timeline)
template <bool CROSS, bool TL>
__global__ __launch_bounds__(DNT) void dec_attn_fwd_kernel(DecAttnArgs p, unsigned long long* tl_out) {
    using G = DecGeo<CROSS>;
    unsigned long long tl[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int i) __attribute__((always_inline)) { if (TL && threadIdx.x == 0) tl[i] = __builtin_readcyclecounter(); };
    stamp(0);
    constexpr int STAGE = G::STAGE, P = G::P, NST = G::NST, REGION = G::REGION, BROWS = G::BROWS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lg = lane >> 4;
    // XCD-aware order (workgroup b runs on XCD b % 8): every XCD gets one contiguous run of (head, sample group) tiles, sample group
    // fastest -- the weight slices of a head (cold: last read a pass ago) are fetched into one or two of the 8 L2s, not into all
    const int ng = (p.B + p.SG - 1) / p.SG;
    int tile_id;
    {
        const int nt = gridDim.x, b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7, loc = b >> 3;
        tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int g = tile_id % ng, h = tile_id / ng;
    const int inner = p.H * 64, T = p.T;
    const int b0 = g * p.SG, sgv = min(p.SG, p.B - b0);            // samples of this workgroup
    const int row0 = b0 * T, nrows = sgv * T, MT = p.B * T;        // its rows of the [B*T, .] matrices
    const int Tk = p.at.Tk;

    // LDS map.  self: [region: ring, later Wo chunks | Q 48 | K 96 | V 96][ctx];  cross: [region: ring, later Wo][V tiles: SG x Tk
    // rows][Q 48][ctx] -- a V tile is read up to its row 63 (probabilities are exactly 0 beyond Tk): whatever follows it must hold
    // finite values
    char* region = smem;
    bf16_t *Qt, *Kt = nullptr, *Vt;
    char* ctxt;
    if (CROSS) {
        Vt = reinterpret_cast<bf16_t*>(smem + REGION);
        Qt = Vt + p.SG * Tk * TS;
        ctxt = reinterpret_cast<char*>(Qt + QROWS * TS);
    } else {
        Qt = reinterpret_cast<bf16_t*>(smem + NCHUNK * CHUNK_BYTES);          // (inside the region, behind the Wo chunks: live after the k-loop)
        Kt = Qt + QROWS * TS;
        Vt = Kt + SELF_KV_ROWS * TS;
        ctxt = smem + REGION;
    }

    // ---- phase 1: the projection tile ------------------------------------------------------------------------------------------
    // per-lane sources of the DMA pieces of a k-tile at k = 0 (LDS image: [A 32 x 64][B BROWS x 64], 16-byte slots XOR-swizzled
    // on the SOURCE side as in gemm_kernel.h)
    const bf16_t* src[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int c = tid + i * DNT;
        if (c < ROWS * 8) {
            const int row = c >> 3, kc = (c & 7) ^ (row & 7);
            src[i] = p.xn + (size_t)min(row0 + row, MT - 1) * p.d + kc * 8;        // rows past the batch repeat its last row
        } else {
            const int c2 = c - ROWS * 8, row = min(c2 >> 3, BROWS - 1), kc = (c2 & 7) ^ (row & 7);
            const int wrow = CROSS ? h * 64 + row : (row >> 6) * inner + h * 64 + (row & 63);
            src[i] = p.w + (size_t)wrow * p.d + kc * 8;
        }
    }
    const int nk = p.d / BK;
    const bool full = wave < 4;                                     // waves 4..7 have no share of the last (half) piece
    auto request = [&](int kt, int s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < P; ++i)
            if (i < P - 1 || full) vlt5gemm::lds_dma16<true>(src[i] + (size_t)kt * BK, region + s * STAGE + (i * DNT + wave * 64) * 16);
    };
    stamp(1);
    // The first two k-tiles are requested before anything else (they are the critical path); the requests that depend on nothing --
    // score addends, cross K fragments and V rows -- follow, then the rest of the ring.  The counted waits of the k-loop stay safe
    // whatever the number of those extra loads: tile kt is always older than the P * (NST-2) youngest requests.
    request(0, 0);
    if (1 < nk) request(1, 1);

    // core roles: wave w runs block w = samples w*spb .. of this workgroup (spb = samples whose rows share a 16-row block)
    const int spb = min(MAXSPB, 16 / T), nblk = (sgv + spb - 1) / spb;
    const int blk_s0 = min(wave, nblk - 1) * spb, nsb = min(spb, sgv - blk_s0);
    const int my_sl = (lrow < nsb * T) ? lrow / T : 0, my_i = (lrow < nsb * T) ? lrow - my_sl * T : 0;
    vlt5attn::AddendRaw raw;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) raw.bv[jb][r] = raw.mv[jb][r] = 0.f;
    if (wave < nblk) vlt5attn::score_addend_load(p.at, b0 + blk_s0 + my_sl, h, my_i - lrow, lane, raw);
    KFromRegs kreg[MAXSPB];
    u32x4_t v0 = {0u, 0u, 0u, 0u}, v1 = v0, v2 = v0, v3 = v0, v4 = v0, v5 = v0;     // (named: an array ends up in scratch)
    if (CROSS) {
#pragma unroll
        for (int s3 = 0; s3 < MAXSPB; ++s3) {
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) kreg[s3].f[jb][ks] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int s3 = 0; s3 < MAXSPB; ++s3) {
            if (wave >= nblk) break;                               // (wave-uniform: only the waves that run a core block hold K)
            const bf16_t* kb = p.at.k + (long long)min(b0 + blk_s0 + s3, p.B - 1) * p.at.k_sb + (long long)h * 64;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    kreg[s3].f[jb][ks] = *reinterpret_cast<const bf16x8_t*>(kb + (long long)min(jb * 16 + lrow, Tk - 1) * p.at.k_st + (ks * 4 + lg) * 8);
        }
        const int nv = p.SG * Tk * 8;                              // 16-byte pieces of the V tiles (<= 6 per thread)
        auto ldv = [&](int it) __attribute__((always_inline)) -> u32x4_t {
            const int idx = min(tid + it * DNT, nv - 1);           // (branch-free: all requests in flight together; the surplus ones repeat the last piece)
            const int r = idx >> 3, s = r / Tk, j = r - s * Tk;
            const int bb = min(b0 + s, p.B - 1);                   // (a sample past the batch: a copy of the last one -- finite)
            return *reinterpret_cast<const u32x4_t*>(p.at.v + bb * p.at.v_sb + (long long)j * p.at.v_st + (long long)h * 64 + (idx & 7) * 8);
        };
        v0 = ldv(0); v1 = ldv(1); v2 = ldv(2); v3 = ldv(3); v4 = ldv(4); v5 = ldv(5);
    }
#pragma unroll
    for (int s = 2; s < NST - 1; ++s)
        if (s < nk) request(s, s);
    stamp(2);

    // zero what is read without anybody writing it (cross: Q rows 32..47 and the ctx tile, contiguous; self: the ctx tile -- its Q / K / V
    // padding rows lie inside the ring and are cleared after the k-loop)
    {
        const uint4 z = make_uint4(0, 0, 0, 0);
        if (CROSS) {
            for (int i = tid; i < ((QROWS - ROWS) * TS * 2 + CTX_BYTES) / 16; i += DNT) reinterpret_cast<uint4*>(Qt + ROWS * TS)[i] = z;
        } else {
            for (int i = tid; i < CTX_BYTES / 16; i += DNT) reinterpret_cast<uint4*>(ctxt)[i] = z;
        }
    }
    if (CROSS) {                                                    // the V rows requested above -> natural tiles
        const int nv = p.SG * Tk * 8;
        auto stv = [&](int it, const u32x4_t& v) __attribute__((always_inline)) {
            const int idx = tid + it * DNT;
            if (idx < nv) *reinterpret_cast<u32x4_t*>(Vt + (idx >> 3) * TS + (idx & 7) * 8) = v;
        };
        // (one unconditional wait for all six: left to the conditional stores, the compiler keeps a load "pending" on the paths that
        // skip a store and guards every later reuse of its registers with a vmcnt(0) -- inside the k-loop)
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5));
        stv(0, v0); stv(1, v1); stv(2, v2); stv(3, v3); stv(4, v4); stv(5, v5);
    }
    const int rb = wave >> 2, cg = wave & 3;                        // 2 row blocks x 4 column groups
    constexpr int NJ = CROSS ? 1 : 3;                               // 16-column fragments per wave
    f32x4_t acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
        // the `young` newest tiles may still be in flight; everything older (tile kt, and the requests above) has landed.  (No
        // requests past the last tile: a quarter of the k-loop's LDS-fill traffic when d_model = 768 is 12 tiles.)
        const int young = min(NST - 2, nk - 1 - kt);
        if (young == NST - 2) { if (full) wait_vm<P * (NST - 2)>(); else wait_vm<(P - 1) * (NST - 2)>(); }
        else if (young == 0) wait_vm<0>();
        else if (young == 1) { if (full) wait_vm<P>(); else wait_vm<P - 1>(); }
        else if (young == 2) { if (full) wait_vm<P * 2>(); else wait_vm<(P - 1) * 2>(); }
        else if (young == 3) { if (full) wait_vm<P * 3>(); else wait_vm<(P - 1) * 3>(); }
        else if (young == 4) { if (full) wait_vm<P * 4>(); else wait_vm<(P - 1) * 4>(); }
        else if (young == 5) { if (full) wait_vm<P * 5>(); else wait_vm<(P - 1) * 5>(); }
        else { if (full) wait_vm<P * 6>(); else wait_vm<(P - 1) * 6>(); }
        __builtin_amdgcn_s_barrier();
        if (TL && kt == 0) stamp(3);
        if (kt + NST - 1 < nk) request(kt + NST - 1, (kt + NST - 1) % NST);     // into the stage tile kt-1 was read from
        const char* at = region + (kt % NST) * STAGE;
        const char* bt = at + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8_t fa = *reinterpret_cast<const bf16x8_t*>(at + lds_off(rb * 16 + lrow, ks * 4 + lg));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(bt + lds_off(cg * 16 * NJ + j * 16 + lrow, ks * 4 + lg));
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, acc[j], 0, 0, 0);
            }
        }
    }
    stamp(4);
    wait_vm<0>();                                                    // the surplus requests have landed: the ring is free
    __builtin_amdgcn_s_waitcnt(0x0F70);                              // (the same wait as the compiler's own instruction: its scoreboard is clear)
    // Everything requested at the top has landed too.  The compiler does not see the assembly DMA requests: a wait it places in front
    // of the first use of those registers, counted in the operations it knows, would also wait for the Wo chunks requested below --
    // so the registers are "used" here, where that wait is free.
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) { asm volatile("" : "+v"(raw.bv[jb][r])); asm volatile("" : "+v"(raw.mv[jb][r])); }
    if (CROSS) {
#pragma unroll
        for (int s3 = 0; s3 < MAXSPB; ++s3)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(kreg[s3].f[jb][ks]));
    }
    __syncthreads();

    // ---- the Wo columns of head h, rows [d], on their way into the retired ring while the core runs ---------------------------
    const int nchunks = (p.d + CHUNK_ROWS - 1) / CHUNK_ROWS;
    auto request_wo = [&](int c0) __attribute__((always_inline)) {   // chunks c0 .. c0+NCHUNK-1 -> region slots 0..NCHUNK-1
        // (all source addresses first, then the requests back to back: address arithmetic between the assembly requests drew a
        // full vmcnt(0) from the compiler's wait-count pass)
        constexpr int PC = CHUNK_ROWS * 8 / DNT;
        const bf16_t* ws[NCHUNK][PC];
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
            for (int i = 0; i < PC; ++i) {
                const int sl = tid + i * DNT, row = sl >> 3, kc = (sl & 7) ^ (row & 7);
                const int n = min((c0 + c) * CHUNK_ROWS + row, p.d - 1);
                ws[c][i] = p.wo + (size_t)n * inner + h * 64 + kc * 8;
            }
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
            if (c0 + c < nchunks) {
#pragma unroll
                for (int i = 0; i < PC; ++i) vlt5gemm::lds_dma16<true>(ws[c][i], region + c * CHUNK_BYTES + (i * DNT + wave * 64) * 16);
            }
        }
    };
    request_wo(0);
    stamp(5);

    // ---- hand-over: accumulators -> bf16 natural tiles (lane: row rb*16 + lrow, 4 consecutive columns) ---------------------------
    if (!CROSS) {                                                    // padding rows of the self-attention tiles (inside the retired ring)
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int i = tid; i < (QROWS - ROWS) * TS * 2 / 16; i += DNT) reinterpret_cast<uint4*>(Qt + ROWS * TS)[i] = z;
        for (int i = tid; i < (SELF_KV_ROWS - ROWS) * TS * 2 / 16; i += DNT) {
            reinterpret_cast<uint4*>(Kt + ROWS * TS)[i] = z;
            reinterpret_cast<uint4*>(Vt + ROWS * TS)[i] = z;
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = cg * 16 * NJ + j * 16;                        // 16 | 64: a fragment lies inside one of q / k / v
        bf16_t* tl_ = CROSS ? Qt : (n < 64 ? Qt : (n < 128 ? Kt : Vt));
        uint2 pk;
        pk.x = pack_bf16x2(acc[j][0], acc[j][1]);
        pk.y = pack_bf16x2(acc[j][2], acc[j][3]);
        *reinterpret_cast<uint2*>(tl_ + (rb * 16 + lrow) * TS + (n & 63) + lg * 4) = pk;
    }
    __syncthreads();
    // the projected rows to HBM (saved for the backward): whole 128-byte rows, 16 bytes per lane
    {
        constexpr int NPART = CROSS ? 1 : 3;
        const int ldq = NPART * inner;
        for (int idx = tid; idx < ROWS * NPART * 8; idx += DNT) {
            const int row = idx / (NPART * 8), rem = idx - row * (NPART * 8), part = rem >> 3, ch = rem & 7;
            if (row < nrows) {
                const bf16_t* tl_ = CROSS ? Qt : (part == 0 ? Qt : (part == 1 ? Kt : Vt));
                const uint4 v = *reinterpret_cast<const uint4*>(tl_ + row * TS + ch * 8);
                *reinterpret_cast<uint4*>(p.qkv + (size_t)(row0 + row) * ldq + part * inner + h * 64 + ch * 8) = v;
            }
        }
    }

    stamp(6);
    // ---- phase 2: attention core, one wave per 16-row block of samples ---------------------------------------------------------------
    if (wave < nblk) {
        float add[4][4];
        vlt5attn::score_addend_finish(p.at, my_i - lrow, lane, raw, add);
        const int brow0 = blk_s0 * T;                                // first row of the block in the workgroup's tiles
        if (CROSS) {
            const bf16_t* const Vs[MAXSPB] = {Vt + (blk_s0 + 0) * Tk * TS, Vt + min(blk_s0 + 1, p.SG - 1) * Tk * TS, Vt + min(blk_s0 + 2, p.SG - 1) * Tk * TS};
            dec_core_block<4>(p.at, h, lane, nsb, T, b0 + blk_s0, Qt + brow0 * TS, kreg, Vs, add, ctxt, brow0);
        } else {
            const vlt5attn::KFromLds kl[MAXSPB] = {{Kt + (brow0 + 0 * T) * TS, lrow, lg}, {Kt + (brow0 + 1 * T) * TS, lrow, lg}, {Kt + (brow0 + 2 * T) * TS, lrow, lg}};
            const bf16_t* const Vs[MAXSPB] = {Vt + (brow0 + 0 * T) * TS, Vt + (brow0 + 1 * T) * TS, Vt + (brow0 + 2 * T) * TS};
            dec_core_block<1>(p.at, h, lane, nsb, T, b0 + blk_s0, Qt + brow0 * TS, kl, Vs, add, ctxt, brow0);
        }
    }

    stamp(7);
    // ---- phase 3: slab h of the output projection ----------------------------------------------------------------------------------
    float* slab = p.slabs + (size_t)h * p.slab_stride;
    const int c = wave >> 1, half = wave & 1;                        // chunk slot and its 96-row half
    for (int c0 = 0; c0 < nchunks; c0 += NCHUNK) {
        wait_vm<0>();                                                // Wo chunks landed (and this wave's ctx / lse stores left)
        if (TL && c0 == 0) stamp(8);
        __syncthreads();                                             // ... everybody's; the ctx tile is complete
        if (TL && c0 == 0) stamp(9);
        if (c0 + c < nchunks) {
            const char* bt = region + c * CHUNK_BYTES;
            f32x4_t o[2][6];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 6; ++j) o[r][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t fa[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) fa[r] = *reinterpret_cast<const bf16x8_t*>(ctxt + lds_off(r * 16 + lrow, ks * 4 + lg));
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(bt + lds_off(half * 96 + j * 16 + lrow, ks * 4 + lg));
#pragma unroll
                    for (int r = 0; r < 2; ++r) o[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa[r], o[r][j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = r * 16 + lrow;
                if (row < nrows) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const int n = (c0 + c) * CHUNK_ROWS + half * 96 + j * 16 + lg * 4;
                        if (n < p.d) vlt5gemm::st16f(slab + (size_t)(row0 + row) * p.d + n, make_float4(o[r][j][0], o[r][j][1], o[r][j][2], o[r][j][3]));
                    }
                }
            }
        }
        if (c0 + NCHUNK < nchunks) {                                 // d_model > 768: the next batch of chunks over the same slots
            __syncthreads();
            request_wo(c0 + NCHUNK);
        }
    }
    if (TL) {
        stamp(10);
        wait_vm<0>();
        stamp(11);
        if (threadIdx.x == 0 && tl_out)
            for (int q = 0; q < 12; ++q) tl_out[(size_t)blockIdx.x * 12 + q] = tl[q];
    }
}

unsigned long long* g_tl_buf = nullptr;      // vlt5dbg_dec_attn_timeline: device buffer of 12 x u64 per workgroup, or null

int check_common(const vlt5_dec_attn_desc* d) {
    if (!d || !d->xn_bf16 || !d->w_bf16 || !d->wo_bf16 || !d->proj_bf16 || !d->o_slabs || !d->core.ctx) return VLT5_ERR_ARG;
    const vlt5_attn_desc& c = d->core;
    if (c.dk != 64 || c.B < 1 || c.H < 1 || c.Tq < 1 || c.Tq > 16 || c.Tk < 1 || c.Tk > 64 || d->d_model < 64) return VLT5_ERR_ARG;
    if ((d->d_model & 63) || (c.o_sb & 3) || (c.o_st & 3) || (((uintptr_t)d->o_slabs) & 15) || (d->slab_stride & 3)) return VLT5_ERR_ALIGN;
    if (d->slab_stride < (long long)c.B * c.Tq * d->d_model) return VLT5_ERR_ARG;
    return VLT5_OK;
}
void fill_core(const vlt5_attn_desc& c, AttnArgs& t) {
    t.q = (const bf16_t*)c.q; t.k = (const bf16_t*)c.k; t.v = (const bf16_t*)c.v;
    t.q_sb = c.q_sb; t.q_st = c.q_st; t.k_sb = c.k_sb; t.k_st = c.k_st; t.v_sb = c.v_sb; t.v_st = c.v_st;
    t.ctx = (bf16_t*)c.ctx; t.o_sb = c.o_sb; t.o_st = c.o_st; t.lse = c.lse;
    t.bias = c.bias; t.bias_q = c.bias_q; t.bias_k = c.bias_k;
    t.key_mask = c.key_mask; t.mask_value = c.mask_value; t.causal = c.causal;
    t.B = c.B; t.H = c.H; t.Tq = c.Tq; t.Tk = c.Tk; t.dk = 64;
    t.drop_thr = c.drop_p > 0.f ? drop_thr16(c.drop_p) : 0u; t.drop_seed = c.drop_seed;
    t.d_ctx = nullptr; t.do_sb = t.do_st = 0; t.dq = t.dk_ = t.dv = nullptr;
    t.dq_sb = t.dq_st = t.dk_sb = t.dk_st = t.dv_sb = t.dv_st = 0; t.dbias = nullptr;
}

template <bool CROSS>
int launch(const vlt5_dec_attn_desc* d, int SG, int lds, void* stream) {
    static std::atomic<unsigned long long> optin{0};
    static std::atomic<unsigned long long> optin_tl{0};
    int rc = vlt5_lds_optin(reinterpret_cast<const void*>(&dec_attn_fwd_kernel<CROSS, false>), LDS_MAX, optin);
    if (rc) return rc;
    rc = vlt5_lds_optin(reinterpret_cast<const void*>(&dec_attn_fwd_kernel<CROSS, true>), LDS_MAX, optin_tl);
    if (rc) return rc;
    DecAttnArgs a;
    a.xn = (const bf16_t*)d->xn_bf16; a.w = (const bf16_t*)d->w_bf16; a.wo = (const bf16_t*)d->wo_bf16; a.qkv = (bf16_t*)d->proj_bf16;
    a.slabs = d->o_slabs; a.slab_stride = d->slab_stride;
    fill_core(d->core, a.at);
    a.B = d->core.B; a.T = d->core.Tq; a.H = d->core.H; a.d = d->d_model; a.SG = SG;
    const dim3 grid(((a.B + SG - 1) / SG) * a.H);
    vlt5gemm::TimingState& tm = vlt5_gemm_timing_state;           // bench.py's in-situ roofline covers these MFMA kernels too
    if (g_tl_buf) hipLaunchKernelGGL((dec_attn_fwd_kernel<CROSS, true>), grid, dim3(DNT), lds, (hipStream_t)stream, a, g_tl_buf);
    else if (tm.on && tm.rec.size() < tm.cap) {
        const size_t i = tm.rec.size();
        vlt5_gemm_timing_rec r;
        const int inner = a.H * 64;
        // flops of the two projections: 2 M (Nproj d + d inner), recorded as one M x N x K problem with the same product
        r.M = a.B * a.T; r.N = (CROSS ? 1 : 3) * inner + inner; r.K = a.d;
        r.batch = 1; r.tile_m = ROWS; r.tile_n = CROSS ? 64 : 192; r.a_kmajor = 0; r.b_kmajor = 0;
        r.splits = 1; r.workgroups = grid.x; r.out_f32 = 1; r.ms = 0.f; r.M2 = 0; r.N2 = 0; r.K2 = 0; r.batch2 = 0;
        tm.rec.push_back(r);
        hipExtLaunchKernelGGL((dec_attn_fwd_kernel<CROSS, false>), grid, dim3(DNT), lds, (hipStream_t)stream, tm.ev[2 * i], tm.ev[2 * i + 1], 0, a,
                              (unsigned long long*)nullptr);
    } else hipLaunchKernelGGL((dec_attn_fwd_kernel<CROSS, false>), grid, dim3(DNT), lds, (hipStream_t)stream, a, (unsigned long long*)nullptr);
    LAUNCH_CHECK();
    return VLT5_OK;
}

}  // namespace

// samples per workgroup of the two kernels (0: the shape is not served -- T > 16)
static int self_sg(int T) { return T > 16 ? 0 : std::min(8, ROWS / T); }
static int cross_sg(int T, int Tk) {
    if (T > 16 || Tk > 64) return 0;
    const int room = (LDS_MAX - CROSS_FIXED_LDS) / (Tk * TS * 2);           // V tiles that fit beside the fixed part
    return std::min(std::min(8, ROWS / T), std::min(room, 6 * DNT / (Tk * 8)));   // (<= 6 sixteen-byte V pieces per thread)
}

// debug (not part of the public ABI): while `buf` (device, 12 x u64 per workgroup) is set, the fused decoder kernels run their
// instrumented build: [0] start, [1] addresses set up, [2] prologue requested, [3] first tile landed, [4] k-loop done, [5] Wo requested,
// [6] hand-over + row stores issued, [7] core done, [8] Wo landed, [9] barrier, [10] slab stores issued, [11] stores drained
extern "C" int vlt5dbg_dec_attn_timeline(void* buf) { g_tl_buf = (unsigned long long*)buf; return VLT5_OK; }

extern "C" int vlt5_dec_attn_fused_ok(int T, int Tk_cross, int d_kv, int d_model) {
    return d_kv == 64 && !(d_model & 63) && d_model >= 64 && self_sg(T) >= 1 && cross_sg(T, Tk_cross) >= 1;
}

// decoder self-attention sublayer between the two norms: q|k|v projection of head h, causal core, slab h of the output projection
extern "C" int vlt5_dec_self_attn_fwd(const vlt5_dec_attn_desc* d, void* stream) {
    int rc = check_common(d);
    if (rc) return rc;
    const vlt5_attn_desc& c = d->core;
    const int inner = c.H * 64, T = c.Tq;
    const bf16_t* base = (const bf16_t*)d->proj_bf16;
    if (c.Tk != T) return VLT5_ERR_ARG;
    if (c.q != (const void*)base || c.k != (const void*)(base + inner) || c.v != (const void*)(base + 2 * inner)) return VLT5_ERR_ARG;
    if (c.q_st != 3 * inner || c.k_st != 3 * inner || c.v_st != 3 * inner) return VLT5_ERR_ARG;
    if (c.q_sb != (long long)T * 3 * inner || c.k_sb != c.q_sb || c.v_sb != c.q_sb) return VLT5_ERR_ARG;
    const int SG = self_sg(T);
    if (SG < 1) return VLT5_ERR_ARG;
    return launch<false>(d, SG, SELF_LDS, stream);
}

// decoder cross-attention sublayer between the two norms: q projection of head h, core over the Tk encoder-side keys / values
// (projected beforehand for all layers: core.k / core.v and their strides), slab h of the output projection
extern "C" int vlt5_cross_attn_fwd(const vlt5_dec_attn_desc* d, void* stream) {
    int rc = check_common(d);
    if (rc) return rc;
    const vlt5_attn_desc& c = d->core;
    const int inner = c.H * 64, T = c.Tq;
    if (!c.k || !c.v || (c.k_sb & 7) || (c.k_st & 7) || (c.v_sb & 7) || (c.v_st & 7)) return VLT5_ERR_ARG;
    if (c.q != d->proj_bf16 || c.q_st != inner || c.q_sb != (long long)T * inner || c.causal) return VLT5_ERR_ARG;
    const int SG = cross_sg(T, c.Tk);
    if (SG < 1) return VLT5_ERR_ARG;
    return launch<true>(d, SG, CROSS_FIXED_LDS + SG * c.Tk * TS * 2, stream);
}
