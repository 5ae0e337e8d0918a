// T5 attention core for short sequences (Tq, Tk <= 64, d_kv <= 64): one workgroup per (batch, head).
//
//   P = softmax(Q K^T + bias + key-mask + causal-mask)   (no 1/sqrt(d): T5)      [fp32]
//   ctx = dropout(P) V
//
// Everything of one (b,h) lives on chip: Q, K, V tiles in LDS (the whole K/V of a head is
// 2 x 64 x 64 bf16 = 16 KB), scores and probabilities in registers.  Four waves, wave w owns query
// rows 16w..16w+15.  MFMAs are issued with swapped operands (keys as the A operand, queries as B)
// so a lane holds S[i = lane&15][j = 16*jb + 4*(lane>>4) + r]: a score row is spread over only
// 4 lanes -> the softmax reductions are 15 in-lane ops + 2 wave shuffles, and the probabilities are
// already in the B-operand layout of the P.V MFMA (the key "slots" of lane group g are
// {32kk + 4g .. +3, 32kk + 16 + 4g .. +3}; V^T is gathered with that slot pattern from the natural V tile by the
// transpose read ds_read_b64_tr_b16), so P never
// leaves registers.  The backward recomputes P from the saved row log-sum-exp and regenerates the
// dropout mask from the counter-based hash.
#include "attn_core.h"
#include <cstdlib>

using namespace vlt5attn;

namespace {

// DK64: d_kv == 64 (every T5 size): the head-dimension loops have constant trip counts and unroll, so all the LDS reads of
// a phase are in flight together (the transpose reads are 8-byte-per-lane reads: latency-bound unless many are queued)
template <bool DK64>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem + TILE_BYTES);
    bf16_t* Vs = reinterpret_cast<bf16_t*>(smem + 2 * TILE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int i0 = wave * 16;
    float add[4][4];
    if (i0 < p.Tq) score_addend(p, b, h, i0, lane, add);
    {
        TileRegs rq, rk, rv;
        tile_load(p.q + b * p.q_sb + (long long)h * p.dk, p.q_st, p.Tq, p.dk, tid, rq);
        tile_load(p.k + b * p.k_sb + (long long)h * p.dk, p.k_st, p.Tk, p.dk, tid, rk);
        tile_load(p.v + b * p.v_sb + (long long)h * p.dk, p.v_st, p.Tk, p.dk, tid, rv);
        tile_store(Qs, tid, rq);
        tile_store(Ks, tid, rk);
        tile_store(Vs, tid, rv);
    }
    __syncthreads();
    if (i0 >= p.Tq) return;
    attn_fwd_rows<DK64>(p, Qs, Ks, Vs, b, h, i0, lane, add);
}

// -DATTN_BWD_NO_STORE (experiment builds only, tools/r05_attn_bwd_fusion_bound.sh): the dq / dk / dv stores of the ENCODER-shaped launches
// (Tq == Tk > 32, not causal) are predicated off at run time -- all the arithmetic stays (the compiler cannot drop it), the 20.6 MB per layer
// never leave the chip: the producer-side upper bound of fusing this kernel with the q|k|v input-gradient GEMM behind it.  Results are WRONG.
#ifdef ATTN_BWD_NO_STORE
#define BWD_STORE_OK(p) (!((p).Tq == (p).Tk && (p).Tq > 32 && !(p).causal && (p).lse != nullptr && (p).H > 0) || (p).B < 0)
#else
#define BWD_STORE_OK(p) true
#endif
// ---- backward building blocks (one wave each) -------------------------------------------------------------------------------
// phase A, query rows i0 .. i0+15: recompute P from the saved log-sum-exp, dP = dO V^T, dS = P (dP - sum_j dP P); leaves
// Pd = dropout(P) and dS in registers (pd, ds; the swapped score layout), stores the dQ rows and the bias-block gradient
template <bool DK64>
__device__ __forceinline__ void attn_bwd_rows(const AttnArgs& p, const bf16_t* Qs, const bf16_t* Ks, const bf16_t* Vs, const bf16_t* dOs,
                                              int b, int h, int i0, int lane, const float (&add)[4][4], float lse, float (&pd)[4][4],
                                              float (&ds)[4][4]) {
    const int lr = lane & 15, g = lane >> 4, i = i0 + lr;
    const long long hoff = (long long)h * p.dk;
    const int ndb = DK64 ? 4 : (p.dk + 15) / 16;
    const int nks = DK64 ? 2 : (p.dk > 32 ? 2 : 1);
    float s[4][4];
    scores_16x64<DK64>(p, Qs, Ks, b, h, i0, lane, add, s);
    // dPd[i][j] = sum_d dO[i][d] V[j][d]  (same swapped layout as the scores)
    f32x4_t dacc[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) dacc[jb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        if (ks >= nks) break;
        bf16x8_t fo = lds_frag(dOs, i0 + lr, ks * 4 + g);
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            bf16x8_t fv = lds_frag(Vs, jb * 16 + lr, ks * 4 + g);
            dacc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fo, dacc[jb], 0, 0, 0);
        }
    }
    const float dsc = drop_scale(p.drop_thr);
    float dsum = 0.f;
    float pr[4][4], dp[4][4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        bool kp[4] = {true, true, true, true};
        if (p.drop_thr)
            drop_keep4(p.drop_seed, (uint32_t)((((size_t)b * p.H + h) * p.Tq + i) * p.Tk + jb * 16 + g * 4), p.drop_thr, kp);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = jb * 16 + g * 4 + r;
            const float pv = (i < p.Tq && j < p.Tk) ? fast_exp(s[jb][r] - lse) : 0.f;
            const float keep = p.drop_thr ? (kp[r] ? dsc : 0.f) : 1.f;
            pr[jb][r] = pv;
            pd[jb][r] = pv * keep;
            dp[jb][r] = dacc[jb][r] * keep;
            dsum += dp[jb][r] * pv;
        }
    }
    dsum = quad_lane_sum(dsum);
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ds[jb][r] = pr[jb][r] * (dp[jb][r] - dsum);
            const int j = jb * 16 + g * 4 + r;
            if (p.dbias && i < p.bias_q && j < p.bias_k && i < p.Tq && j < p.Tk)
                p.dbias[(((size_t)b * p.H + h) * p.bias_q + i) * p.bias_k + j] = ds[jb][r];
        }
    // dQ[i][d] = sum_j dS[i][j] K[j][d]   (K^T gathered from the natural K tile with transpose reads)
    bf16x8_t dsf[2] = {pack_slots(ds[0], ds[1]), pack_slots(ds[2], ds[3])};
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        if (db >= ndb) break;
        f32x4_t o = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk * 32 < p.Tk) {
                bf16x8_t fk = frag_tr_slots(Ks, db * 16, kk, lane);
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, dsf[kk], o, 0, 0, 0);
            }
        }
        const int d = db * 16 + g * 4;
        if (i < p.Tq && d < p.dk && BWD_STORE_OK(p)) {
            uint2 pk;
            pk.x = pack_bf16x2(o[0], o[1]);
            pk.y = pack_bf16x2(o[2], o[3]);
            *reinterpret_cast<uint2*>(p.dq + b * p.dq_sb + (long long)i * p.dq_st + hoff + d) = pk;
        }
    }
}
// Pd / dS of rows i0 .. i0+15 into the natural [i][j] tiles: 4 consecutive j per (lane, jb) -> one 8-byte store each
__device__ __forceinline__ void attn_bwd_put(bf16_t* Ps, bf16_t* dSs, int i0, int lane, const float (&pd)[4][4], const float (&ds)[4][4]) {
    const int lr = lane & 15, g = lane >> 4, i = i0 + lr;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        uint2 a, c;
        a.x = pack_bf16x2(pd[jb][0], pd[jb][1]); a.y = pack_bf16x2(pd[jb][2], pd[jb][3]);
        c.x = pack_bf16x2(ds[jb][0], ds[jb][1]); c.y = pack_bf16x2(ds[jb][2], ds[jb][3]);
        *reinterpret_cast<uint2*>(Ps + i * TS + jb * 16 + g * 4) = a;
        *reinterpret_cast<uint2*>(dSs + i * TS + jb * 16 + g * 4) = c;
    }
}
// phase B, key rows j0 .. j0+15:  dV = Pd^T dO,  dK = dS^T Q  (reduction over the queries i); all four operands are transposed
// views of natural tiles -> ds_read_b64_tr_b16
template <bool DK64>
__device__ __forceinline__ void attn_bwd_keys(const AttnArgs& p, const bf16_t* Ps, const bf16_t* dSs, const bf16_t* dOs, const bf16_t* Qs,
                                              int b, int h, int j0, int lane) {
    const int lr = lane & 15, g = lane >> 4, j = j0 + lr;
    const long long hoff = (long long)h * p.dk;
    const int ndb = DK64 ? 4 : (p.dk + 15) / 16;
    const int nis = p.Tq > 32 ? 2 : 1;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        if (db >= ndb) break;
        f32x4_t av = (f32x4_t){0.f, 0.f, 0.f, 0.f}, ak = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks >= nis) break;
            bf16x8_t fp = frag_tr_std(Ps, j0, ks, lane);          // Pd^T[j][i]
            bf16x8_t fs = frag_tr_std(dSs, j0, ks, lane);         // dS^T[j][i]
            bf16x8_t fo = frag_tr_std(dOs, db * 16, ks, lane);    // dO^T[d][i]
            bf16x8_t fq = frag_tr_std(Qs, db * 16, ks, lane);     // Q^T[d][i]
            av = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fo, fp, av, 0, 0, 0);
            ak = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq, fs, ak, 0, 0, 0);
        }
        const int d = db * 16 + g * 4;
        if (j < p.Tk && d < p.dk && BWD_STORE_OK(p)) {
            uint2 pk;
            pk.x = pack_bf16x2(av[0], av[1]);
            pk.y = pack_bf16x2(av[2], av[3]);
            *reinterpret_cast<uint2*>(p.dv + b * p.dv_sb + (long long)j * p.dv_st + hoff + d) = pk;
            pk.x = pack_bf16x2(ak[0], ak[1]);
            pk.y = pack_bf16x2(ak[2], ak[3]);
            *reinterpret_cast<uint2*>(p.dk_ + b * p.dk_sb + (long long)j * p.dk_st + hoff + d) = pk;
        }
    }
}

// -DATTN_TIMELINE (debug builds, tools/attn_bwd_timeline.py): wave 0 of every workgroup stamps the shader clock at the phase boundaries.
// (Round 4, measured and NOT kept: the addend words as 16-byte loads, dbias as 16-byte stores and every load of the workgroup requested
// before the first wait take this kernel from 43.9 k to 32.6 k cycles per workgroup at the encoder shape when it is launched alone on
// warm buffers -- and from 14.3 to 14.9-15.1 us per launch inside the step, where its operands come cold from HBM and 960 small
// workgroups already hide the round trips; the extra live registers cost more.  profiles/r04_j_attn_bwd_timeline.txt)
#ifdef ATTN_TIMELINE
static __device__ long long* g_attn_tl = nullptr;
extern "C" int vlt5dbg_set_attn_timeline(long long* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_tl), &buf, sizeof buf); }
#define ATL(i) do { if (g_attn_tl && threadIdx.x == 0) g_attn_tl[(size_t)blockIdx.x * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define ATL(i) do { } while (0)
#endif
template <bool DK64>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ATL(0);
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem + TILE_BYTES);
    bf16_t* Vs = reinterpret_cast<bf16_t*>(smem + 2 * TILE_BYTES);
    bf16_t* dOs = reinterpret_cast<bf16_t*>(smem + 3 * TILE_BYTES);
    bf16_t* Ps = Ks;      // Pd[i][j] (natural), reuses the K tile after phase A
    bf16_t* dSs = Vs;     // dS[i][j] (natural), reuses the V tile after phase A
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const long long hoff = (long long)h * p.dk;
    float add[4][4];
    float lse_row = 0.f;
    if (wave * 16 < p.Tq) {
        score_addend(p, b, h, wave * 16, lane, add);
        const int ir = wave * 16 + (lane & 15);
        if (ir < p.Tq) lse_row = p.lse[((size_t)b * p.H + h) * p.Tq + ir];
    }
    {
        TileRegs rq, rk, rv, ro;
        tile_load(p.q + b * p.q_sb + hoff, p.q_st, p.Tq, p.dk, tid, rq);
        tile_load(p.k + b * p.k_sb + hoff, p.k_st, p.Tk, p.dk, tid, rk);
        tile_load(p.v + b * p.v_sb + hoff, p.v_st, p.Tk, p.dk, tid, rv);
        tile_load(p.d_ctx + b * p.do_sb + hoff, p.do_st, p.Tq, p.dk, tid, ro);
        ATL(1);
        tile_store(Qs, tid, rq);
        tile_store(Ks, tid, rk);
        tile_store(Vs, tid, rv);
        tile_store(dOs, tid, ro);
    }
    __syncthreads();
    ATL(2);
    const int i0 = wave * 16;
    float pd[4][4], ds[4][4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) { pd[jb][r] = 0.f; ds[jb][r] = 0.f; }
    if (i0 < p.Tq) attn_bwd_rows<DK64>(p, Qs, Ks, Vs, dOs, b, h, i0, lane, add, lse_row, pd, ds);      // wave-uniform
    ATL(3);
    __syncthreads();                     // every wave is done with the K / V tiles -> reuse them for Pd / dS
    attn_bwd_put(Ps, dSs, i0, lane, pd, ds);
    __syncthreads();
    ATL(4);
    if (wave * 16 < p.Tk) attn_bwd_keys<DK64>(p, Ps, dSs, dOs, Qs, b, h, wave * 16, lane);              // wave w owns key rows 16w..
    ATL(5);
#ifdef ATTN_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATL(6);
#endif
}

// (Tried and removed, round 2: few-query variants for the decoder (Tq <= 16) in which a WAVE owns a (batch, head), stages its own
// wave-private tiles and runs these building blocks with no workgroup barrier.  Correct, but slower in situ than the kernels here
// -- forward 6.3 / 9.3 us against 6.1 / 8.3 (self / cross), backward 10.3 / 16.9 against 8.8 / 14.7: staging a tile with 64 lanes
// instead of 256 costs more than the three idle waves it removes.)

int fill_args(const vlt5_attn_desc* d, AttnArgs& a, bool bwd) {
    if (!d || !d->q || !d->k || !d->v) return VLT5_ERR_ARG;
    if (d->Tq < 1 || d->Tq > 64 || d->Tk < 1 || d->Tk > 64 || d->dk < 8 || d->dk > 64 || d->B < 1 || d->H < 1) return VLT5_ERR_ARG;
    if (d->dk & 7) return VLT5_ERR_ALIGN;
    long long strides[] = {d->q_sb, d->q_st, d->k_sb, d->k_st, d->v_sb, d->v_st};
    for (long long s : strides) if (s & 7) return VLT5_ERR_ALIGN;
    a.q = (const bf16_t*)d->q; a.k = (const bf16_t*)d->k; a.v = (const bf16_t*)d->v;
    a.q_sb = d->q_sb; a.q_st = d->q_st; a.k_sb = d->k_sb; a.k_st = d->k_st; a.v_sb = d->v_sb; a.v_st = d->v_st;
    a.ctx = (bf16_t*)d->ctx; a.o_sb = d->o_sb; a.o_st = d->o_st; a.lse = d->lse;
    a.bias = d->bias; a.bias_q = d->bias_q; a.bias_k = d->bias_k;
    a.key_mask = d->key_mask; a.mask_value = d->mask_value; a.causal = d->causal;
    a.B = d->B; a.H = d->H; a.Tq = d->Tq; a.Tk = d->Tk; a.dk = d->dk;
    a.drop_thr = d->drop_p > 0.f ? drop_thr16(d->drop_p) : 0u; a.drop_seed = d->drop_seed;
    a.d_ctx = (const bf16_t*)d->d_ctx; a.do_sb = d->do_sb; a.do_st = d->do_st;
    a.dq = (bf16_t*)d->dq; a.dk_ = (bf16_t*)d->dk_; a.dv = (bf16_t*)d->dv;
    a.dq_sb = d->dq_sb; a.dq_st = d->dq_st; a.dk_sb = d->dk_sb; a.dk_st = d->dk_st; a.dv_sb = d->dv_sb; a.dv_st = d->dv_st;
    a.dbias = d->dbias;
    if (!bwd) {
        if (!d->ctx || (d->o_sb & 3) || (d->o_st & 3)) return VLT5_ERR_ARG;
    } else {
        if (!d->d_ctx || !d->dq || !d->dk_ || !d->dv || !d->lse) return VLT5_ERR_ARG;
        long long s2[] = {d->do_sb, d->do_st, d->dq_sb, d->dq_st, d->dk_sb, d->dk_st, d->dv_sb, d->dv_st};
        for (long long s : s2) if (s & 7) return VLT5_ERR_ALIGN;
    }
    return VLT5_OK;
}

}  // namespace

extern "C" int vlt5_attn_fwd(const vlt5_attn_desc* d, void* stream) {
    AttnArgs a;
    int rc = fill_args(d, a, false);
    if (rc) return rc;
    if (a.dk == 64) hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3(a.B * a.H), dim3(256), 3 * TILE_BYTES, (hipStream_t)stream, a);
    else            hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3(a.B * a.H), dim3(256), 3 * TILE_BYTES, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return VLT5_OK;
}

extern "C" int vlt5_attn_bwd(const vlt5_attn_desc* d, void* stream) {
    AttnArgs a;
    int rc = fill_args(d, a, true);
    if (rc) return rc;
    if (a.dk == 64) hipLaunchKernelGGL(attn_bwd_kernel<true>, dim3(a.B * a.H), dim3(256), 4 * TILE_BYTES, (hipStream_t)stream, a);
    else            hipLaunchKernelGGL(attn_bwd_kernel<false>, dim3(a.B * a.H), dim3(256), 4 * TILE_BYTES, (hipStream_t)stream, a);
    LAUNCH_CHECK();
    return VLT5_OK;
}

int vlt5_build_flags_attn() {
    int f = 0;
#ifdef ATTN_BWD_NO_STORE
    f |= VLT5_BUILD_ATTN_BWD_NO_STORE;
#endif
#ifdef ATTN_TIMELINE
    f |= VLT5_BUILD_TIMELINE;
#endif
    return f;
}
