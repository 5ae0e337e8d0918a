// Device-side building blocks of the T5 attention core (Tq, Tk <= 64, d_kv <= 64), shared by the stand-alone kernels (attn.hip)
// and the fused encoder attention kernels (enc_attn.hip): natural [64][TS] LDS tiles, swapped-operand score MFMAs, in-register
// softmax over 4-lane groups, transposed fragments through ds_read_b64_tr_b16.  Layout notes at the top of attn.hip.
#pragma once
#include "common.h"
#include "vlt5_hip.h"

namespace vlt5attn {

constexpr int TS = 72;                 // LDS row stride in elements (64 + 8 pad) -> 144 B, 16-byte aligned rows
constexpr int TILE_BYTES = 64 * TS * 2;

struct AttnArgs {
    const bf16_t *q, *k, *v;
    long long q_sb, q_st, k_sb, k_st, v_sb, v_st;
    bf16_t* ctx; long long o_sb, o_st;
    float* lse;
    const float* bias; int bias_q, bias_k;
    const float* key_mask; float mask_value;
    int causal;
    int B, H, Tq, Tk, dk;
    uint32_t drop_thr, drop_seed;
    const bf16_t* d_ctx; long long do_sb, do_st;
    bf16_t *dq, *dk_, *dv; long long dq_sb, dq_st, dk_sb, dk_st, dv_sb, dv_st;
    float* dbias;
};

// stage a [T x dk] bf16 matrix (row stride st) into a natural [64][TS] LDS tile (zero filled outside T x dk)
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ src, long long st, int T, int dk, bf16_t* nat, int tid) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        int c = tid + it * 256;
        int row = c >> 3, dc = c & 7;
        uint4 z = make_uint4(0, 0, 0, 0);
        if (row < T && dc * 8 < dk) z = *reinterpret_cast<const uint4*>(src + (long long)row * st + dc * 8);
        *reinterpret_cast<uint4*>(nat + row * TS + dc * 8) = z;
    }
}

// the same in two halves, so that the loads of ALL tiles of a workgroup are in flight together before the first LDS store waits
struct TileRegs { uint4 v[2]; };
__device__ __forceinline__ void tile_load(const bf16_t* __restrict__ src, long long st, int T, int dk, int tid, TileRegs& r) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        int c = tid + it * 256;
        int row = c >> 3, dc = c & 7;
        uint4 z = make_uint4(0, 0, 0, 0);
        if (row < T && dc * 8 < dk) z = *reinterpret_cast<const uint4*>(src + (long long)row * st + dc * 8);
        r.v[it] = z;
    }
}
__device__ __forceinline__ void tile_store(bf16_t* nat, int tid, const TileRegs& r) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        int c = tid + it * 256;
        *reinterpret_cast<uint4*>(nat + (c >> 3) * TS + (c & 7) * 8) = r.v[it];
    }
}

__device__ __forceinline__ bf16x8_t lds_frag(const bf16_t* tile, int row, int chunk) {     // 8 consecutive elements of a row
    return *reinterpret_cast<const bf16x8_t*>(tile + row * TS + chunk * 8);
}
// Transposed fragments straight from a NATURAL tile T[k][r] with the gfx950 transpose read (ds_read_b64_tr_b16): within a
// 16-lane group, lane i points at T[k0 + (i>>2)][r0 + 4*(i&3)] and receives T[k0 .. k0+3][r0 + i].
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ s16x4_t tr4(const bf16_t* tile, int k0, int r0, int lane) {
    const int i = lane & 15;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(tile + (k0 + (i >> 2)) * TS + r0 + 4 * (i & 3)));
}
__device__ __forceinline__ bf16x8_t join8(s16x4_t lo, s16x4_t hi) {
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
// rows r0..r0+15 of T^T, k-slots in the "P slot" pattern of lane group g: {32kk+4g .. +3, 32kk+16+4g .. +3}
__device__ __forceinline__ bf16x8_t frag_tr_slots(const bf16_t* tile, int r0, int kk, int lane) {
    const int g = lane >> 4;
    return join8(tr4(tile, kk * 32 + 4 * g, r0, lane), tr4(tile, kk * 32 + 16 + 4 * g, r0, lane));
}
// rows r0..r0+15 of T^T, k = 32*ks + 8g .. +7 (the standard MFMA operand order)
__device__ __forceinline__ bf16x8_t frag_tr_std(const bf16_t* tile, int r0, int ks, int lane) {
    const int g = lane >> 4;
    return join8(tr4(tile, ks * 32 + 8 * g, r0, lane), tr4(tile, ks * 32 + 8 * g + 4, r0, lane));
}
__device__ __forceinline__ bf16x8_t pack_slots(const float (&lo)[4], const float (&hi)[4]) {
    union { uint32_t u[4]; bf16x8_t v; } r;
    r.u[0] = pack_bf16x2(lo[0], lo[1]); r.u[1] = pack_bf16x2(lo[2], lo[3]);
    r.u[2] = pack_bf16x2(hi[0], hi[1]); r.u[3] = pack_bf16x2(hi[2], hi[3]);
    return r.v;
}

// what is added to the raw scores of this lane's 16 elements: relative-position bias + key mask + causal mask, -inf beyond Tk.
// Depends on nothing in LDS, so it is requested BEFORE the tiles are staged: its global-memory round trip overlaps the staging
// loads instead of following the Q.K^T MFMAs (one round trip less on the critical path of a latency-bound kernel).
struct AddendRaw { float bv[4][4], mv[4][4]; };       // the loaded bias / key-mask words of a 16-row block, before they are combined
// request: every load is issued branch-free per lane (clamped index, selected in score_addend_finish), so the 32 loads of a block are
// in flight together behind ONE wait -- per-element `if (..) v += load` chains serialise into 32 dependent round trips when few
// waves share a CU (measured in the fused kernel: 20 k cycles).  An absent operand reads a dummy word of q instead of branching
// around its loads: one basic block, one wait.
__device__ __forceinline__ void score_addend_load(const AttnArgs& p, int b, int h, int i0, int lane, AddendRaw& raw) {
    const int lr = lane & 15, g = lane >> 4, i = i0 + lr;
    const bool hb = p.bias != nullptr, hm = p.key_mask != nullptr;            // wave-uniform
    const float* row = hb ? p.bias + ((size_t)h * p.bias_q + min(i, p.bias_q - 1)) * p.bias_k : reinterpret_cast<const float*>(p.q);
    const float* mrow = hm ? p.key_mask + (size_t)b * p.Tk : reinterpret_cast<const float*>(p.q);
    const int bk1 = hb ? p.bias_k - 1 : 0, mk1 = hm ? p.Tk - 1 : 0;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            raw.bv[jb][r] = row[min(jb * 16 + g * 4 + r, bk1)];
            raw.mv[jb][r] = mrow[min(jb * 16 + g * 4 + r, mk1)];
        }
}
__device__ __forceinline__ void score_addend_finish(const AttnArgs& p, int i0, int lane, const AddendRaw& raw, float (&add)[4][4]) {
    const int lr = lane & 15, g = lane >> 4, i = i0 + lr;
    const bool hb = p.bias != nullptr, hm = p.key_mask != nullptr;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = jb * 16 + g * 4 + r;
            float v = 0.f;
            if (hb && i < p.bias_q && j < p.bias_k) v += raw.bv[jb][r];
            if (hm) v += (1.0f - raw.mv[jb][r]) * p.mask_value;
            if (p.causal && j > i) v += -10000.0f;
            add[jb][r] = (j < p.Tk) ? v : -INFINITY;
        }
}
__device__ __forceinline__ void score_addend(const AttnArgs& p, int b, int h, int i0, int lane, float (&add)[4][4]) {
    AddendRaw raw;
    score_addend_load(p, b, h, i0, lane, raw);
    score_addend_finish(p, i0, lane, raw, add);
}

// scores of this wave's 16 query rows against all 64 key slots, + the addend above; s[jb][r] in the swapped layout
template <bool DK64>
__device__ __forceinline__ void scores_16x64(const AttnArgs& p, const bf16_t* Qs, const bf16_t* Ks, int b, int h, int i0,
                                             int lane, const float (&add)[4][4], float (&s)[4][4]) {
    const int lr = lane & 15, g = lane >> 4;
    f32x4_t acc[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) acc[jb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int nks = DK64 ? 2 : (p.dk > 32 ? 2 : 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        if (ks >= nks) break;
        bf16x8_t fq = lds_frag(Qs, i0 + lr, ks * 4 + g);
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            bf16x8_t fk = lds_frag(Ks, jb * 16 + lr, ks * 4 + g);
            acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq, acc[jb], 0, 0, 0);
        }
    }
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[jb][r] = (add[jb][r] == -INFINITY) ? -INFINITY : acc[jb][r] + add[jb][r];
}

__device__ __forceinline__ float quad_lane_sum(float v) {     // the 4 lanes sharing a query row: l, l^16, l^32, l^48
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float quad_lane_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}


// one wave, NB consecutive 16-row query blocks (rows i0 .. i0 + 16*NB - 1) of head (b, h): scores -> softmax (+ saved log-sum-exp)
// -> dropout -> P.V -> ctx rows.  Q/K/V are the staged natural tiles of that head, add[n] the score addend of block n.  The blocks
// share every K and V^T fragment read and are independent MFMA / VALU chains in one basic block (the fused encoder kernel runs
// NB = 2: half the LDS reads per row and two chains to interleave); per block the arithmetic and its order are those of NB = 1.
struct NoStamp { __device__ __forceinline__ void operator()(int) const {} };
// where the K fragments come from (default: the staged natural tile) and an optional second destination of the ctx rows (the fused
// decoder kernels keep them in LDS as the operand of the output projection)
struct KFromLds {
    const bf16_t* Ks; int lr, g;
    __device__ __forceinline__ bf16x8_t operator()(int jb, int ks) const { return lds_frag(Ks, jb * 16 + lr, ks * 4 + g); }
};
struct NoSink { __device__ __forceinline__ void operator()(int, int, uint2) const {} };
template <bool DK64, int NB, class KF, class Sink, class Stamp = NoStamp>
__device__ __forceinline__ void attn_fwd_blocks_ex(const AttnArgs& p, const bf16_t* Qs, const KF kf, const bf16_t* Vs, int b, int h,
                                                   int i0, int lane, const float (&add)[NB][4][4], const Sink sink, Stamp stamp = Stamp()) {
    const int lr = lane & 15, g = lane >> 4;
    f32x4_t acc[NB][4];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[n][jb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int nks = DK64 ? 2 : (p.dk > 32 ? 2 : 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        if (ks >= nks) break;
        bf16x8_t fk[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) fk[jb] = kf(jb, ks);
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const bf16x8_t fq = lds_frag(Qs, i0 + n * 16 + lr, ks * 4 + g);
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) acc[n][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[jb], fq, acc[n][jb], 0, 0, 0);
        }
    }
    stamp(0);
    // The softmax of a block is VALU work of all four lanes of a row: kept lean (in the fused encoder kernel it runs with nothing to
    // hide behind).  exp(s - m) = exp2(fma(s, log2e, -m log2e)); the normaliser and the dropout scale are one factor; a masked slot
    // (addend -inf: key >= Tk) needs no select since acc is finite (padding rows hold zeros or copies of valid rows).
    const float dsc = drop_scale(p.drop_thr);          // 1.0 exactly without dropout
    constexpr float kLog2e = 1.4426950408889634f;
    bf16x8_t pf[NB][2];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int i = i0 + n * 16 + lr;
        const uint32_t rowbase = ((uint32_t)(b * p.H + h) * (uint32_t)p.Tq + (uint32_t)i) * (uint32_t)p.Tk + (uint32_t)(g * 4);   // low 32 bits of the element index
        float s[4][4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[jb][r] = acc[n][jb][r] + add[n][jb][r];
        // (reductions as trees over the four key blocks: serial 16-long max / add chains leave a wave with two independent
        // instruction streams waiting on VALU latency)
        float mj[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) mj[jb] = fmaxf(fmaxf(s[jb][0], s[jb][1]), fmaxf(s[jb][2], s[jb][3]));
        const float m = quad_lane_max(fmaxf(fmaxf(mj[0], mj[1]), fmaxf(mj[2], mj[3])));
        const float m2 = m * kLog2e;
        float sj[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s[jb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[jb][r], kLog2e, -m2));
            sj[jb] = (s[jb][0] + s[jb][1]) + (s[jb][2] + s[jb][3]);
        }
        const float sum = quad_lane_sum((sj[0] + sj[1]) + (sj[2] + sj[3]));
        const float vs = __builtin_amdgcn_rcpf(sum) * dsc;
        if (g == 0 && i < p.Tq && p.lse) p.lse[((size_t)b * p.H + h) * p.Tq + i] = m + logf(sum);
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            bool keep[4] = {true, true, true, true};
            if (p.drop_thr) drop_keep4(p.drop_seed, rowbase + jb * 16, p.drop_thr, keep);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[jb][r] = keep[r] ? s[jb][r] * vs : 0.f;
        }
        pf[n][0] = pack_slots(s[0], s[1]);
        pf[n][1] = pack_slots(s[2], s[3]);
    }
    stamp(1);
    const int ndb = DK64 ? 4 : (p.dk + 15) / 16;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        if (db >= ndb) break;
        f32x4_t o[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) o[n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk * 32 < p.Tk) {
                const bf16x8_t fv = frag_tr_slots(Vs, db * 16, kk, lane);        // V^T[d][j] gathered from the natural V tile
#pragma unroll
                for (int n = 0; n < NB; ++n) o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, pf[n][kk], o[n], 0, 0, 0);
            }
        }
        const int d = db * 16 + g * 4;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int i = i0 + n * 16 + lr;
            if (i < p.Tq && d < p.dk) {
                uint2 pk;
                pk.x = pack_bf16x2(o[n][0], o[n][1]);
                pk.y = pack_bf16x2(o[n][2], o[n][3]);
                *reinterpret_cast<uint2*>(p.ctx + b * p.o_sb + (long long)i * p.o_st + (long long)h * p.dk + d) = pk;
                sink(i, d, pk);
            }
        }
    }
}
template <bool DK64, int NB, class Stamp = NoStamp>
__device__ __forceinline__ void attn_fwd_blocks(const AttnArgs& p, const bf16_t* Qs, const bf16_t* Ks, const bf16_t* Vs, int b, int h,
                                                int i0, int lane, const float (&add)[NB][4][4], Stamp stamp = Stamp()) {
    attn_fwd_blocks_ex<DK64, NB>(p, Qs, KFromLds{Ks, lane & 15, lane >> 4}, Vs, b, h, i0, lane, add, NoSink(), stamp);
}
template <bool DK64>
__device__ __forceinline__ void attn_fwd_rows(const AttnArgs& p, const bf16_t* Qs, const bf16_t* Ks, const bf16_t* Vs, int b, int h,
                                              int i0, int lane, const float (&add)[4][4]) {
    float a1[1][4][4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) a1[0][jb][r] = add[jb][r];
    attn_fwd_blocks<DK64, 1>(p, Qs, Ks, Vs, b, h, i0, lane, a1);
}

}  // namespace vlt5attn
