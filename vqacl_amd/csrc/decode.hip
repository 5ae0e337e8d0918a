// Greedy decoding, one token per step on B rows: the kernels of vlt5_decoder_step.
//
// Reference being replaced: HF generate -> VLT5.forward(decoder_input_ids[:, -1:], past_key_values) (VL-T5/src/vqa_model.py:68-121,
// src/modeling_t5_our.py:608-629, 715-772), i.e. per token and decoder layer: T5LayerNorm -> q/k/v Linear (k, v appended to the cache)
// -> softmax(q K^T + bias) V over the cached keys -> o Linear + residual -> T5LayerNorm -> q Linear -> attention over the 58
// encoder-side keys -> o Linear + residual -> T5LayerNorm -> wi + ReLU -> wo + residual; then final norm, rescale, tied lm_head and
// argmax over the vocabulary.
//
// On B = 80 rows every projection is a weight-streaming problem (a layer is 14 MB of bf16 weights for 80 x 7 MFLOP), and a token is
// ~100 dependent launches, so each kernel is built for ONE memory round trip instead of a k-loop:
//   declin_kernel   out[16 rows x 16*NFRAG columns] per workgroup, the reduction split over the waves of the workgroup: every wave
//                   requests its whole share of the weight slice and of the 16 activation rows (fragment-shaped, straight into
//                   VGPRs: each weight byte is used by exactly one wave, an LDS round trip would only add latency) in one burst,
//                   then runs KS x NFRAG MFMAs; partial tiles meet in LDS.  The T5 RMS norm in front of a projection is folded in
//                   (operand = bf16(x * w_norm) built in registers from the f32 residual stream, rows scaled by rstd in the epilogue);
//                   residual add, ReLU, routing of the k | v columns into the cache slot of the step, per-tile argmax of the
//                   vocabulary projection are epilogue options.  The workgroups of one column tile (the 5 row blocks of B = 80)
//                   sit on one XCD, so a weight byte crosses the fabric once.
//   dec_core_kernel one wave per (sample, head): all keys and values of the head requested at once (<= 64 keys), scores, softmax
//                   and P V in f32 on the VALU, cross-lane sums by butterfly exchange -- no LDS, no barrier.
//   dec_io_kernel   between two steps: finishes the argmax (first maximum wins, torch.argmax semantics), applies HF's greedy
//                   bookkeeping (pad after EOS, done flags), gathers the next input embedding row and the relative-position bias row
//                   of the next position.
#include "common.h"
#include "decode.h"

#ifndef DECLIN_WT
#define DECLIN_WT 0                // 1: outputs written through the L2 (sc0 sc1) as the GEMM does; measured 1 % slower on these 245 KB
#endif                             // outputs than plain stores (profiles/r04_d_ab_decode.txt), so plain is the default

// Cache policy of the LDS-DMA / loads (aux 2 = nt).  Same-box A/B at B = 80 (profiles/r04_h_ab_decode_nt.txt): weight stream nt 0.575 against
// 0.601 ms per token-step (a slice is used by the 5 row blocks of one XCD within microseconds and by nobody afterwards: it should not
// displace the activations and the cache in L2); nt on the bf16 activation rows (re-read by all 48 column tiles) 0.611, on the keys /
// values of the attention core 0.597 -- both left at the default policy.
#ifndef DECLIN_W_AUX
#define DECLIN_W_AUX 2
#endif
#ifndef DECLIN_X_AUX
#define DECLIN_X_AUX 0
#endif
#ifndef DEC_CORE_NT
#define DEC_CORE_NT 0
#endif
#ifndef DECLIN_VOCAB
#define DECLIN_VOCAB 1             // 0: the vocabulary projection through declin_rows_kernel (A/B builds)
#endif
#ifndef DECLIN_ROWS_W_AUX
#define DECLIN_ROWS_W_AUX DECLIN_W_AUX     // ... of the vocabulary projection (its weight is read once chip-wide)
#endif

#ifdef DECLIN_TIMELINE
#define TLS(i) do { if (a.tl && threadIdx.x == 0) a.tl[(size_t)blockIdx.x * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TLS(i) do { } while (0)
#endif

namespace {

__device__ __forceinline__ void dl_store16f(float* dst, float4 v) {
#if DECLIN_WT
    store_wt16f(dst, v);
#else
    *reinterpret_cast<float4*>(dst) = v;
#endif
}
__device__ __forceinline__ void dl_store8(void* dst, uint2 v) {
#if DECLIN_WT
    store_wt8(dst, v);
#else
    *reinterpret_cast<uint2*>(dst) = v;
#endif
}

// ------------------------------------------------------------------------------------------------------------------------------
// declin: out[m, n] = epi(rowscale[m] * alpha * sum_k A[m, k] W[n, k])
// ------------------------------------------------------------------------------------------------------------------------------
// Operand staging.  Fragment-shaped loads straight into VGPRs (16 rows x 64 bytes per wave instruction) kept the texture-address unit
// busy 52 cycles per instruction (PMC: TA_BUSY = half the kernel, 64 cache accesses per instruction; profiles/r04_f_decode_pmc.txt),
// i.e. ~17 bytes per clock and CU -- the kernel was bound by that, not by HBM.  So the weight slice (and a bf16 activation operand)
// now arrive by LDS-DMA in whole 128-byte lines: a piece = 8 rows x 128 bytes per wave instruction, lane l fetches chunk
// (l & 7) ^ (row & 7) of row l >> 3 into the linear LDS slot l, so that a row's chunk c sits in slot c ^ (row & 7) and the
// ds_read_b128 of a fragment (16 rows, same chunk) spreads over the banks.  Every wave stages only ITS k-range (the reduction is still
// split over the waves): no workgroup barrier before the MFMAs, one counted wait per wave.  An f32 activation operand (norm folded in)
// stays on the register path: its values are multiplied by the norm weight and squared on the way.
typedef __attribute__((address_space(3))) void* dl_lds_ptr_t;

template <bool AF32, int KT, int NFRAG>
__global__ __launch_bounds__(512) void declin_kernel(const DecLinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dl_smem[];
    TLS(0);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, NW = blockDim.x >> 6;
    // workgroup -> (column tile, row block): the row blocks of a column tile share an XCD (block b runs on XCD b % 8), so the tile's
    // weight slice is fetched into one L2 once
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int ct = xcd * a.ct_per_xcd + j / a.RB, rb = j % a.RB;
    if (ct >= a.CT) return;
    const int r16 = lane & 15, kq = lane >> 4;
    const int m = rb * 16 + r16;
    const int mc = m < a.rows ? m : a.rows - 1;                  // rows past the end read the last row (never stored)
    const int n_tile = ct * (16 * NFRAG);
    // LDS carve: [per wave: weight tiles KT x (NFRAG*16 rows x 128 B) | bf16 activation tiles KT x (16 x 128 B)] | partial tiles |
    // norm-weight strips | partial sums of squares | argmax exchange
    constexpr int WT_BYTES = NFRAG * 16 * 128, XT_BYTES = AF32 ? 0 : 16 * 128, WAVE_BYTES = KT * (WT_BYTES + XT_BYTES);
    char* stage = dl_smem + (size_t)w * WAVE_BYTES;
    float4* red = reinterpret_cast<float4*>(dl_smem + (size_t)NW * WAVE_BYTES);           // [NW][NFRAG][64]
    float* wst = reinterpret_cast<float*>(red + (size_t)NW * NFRAG * 64);                 // [NW][256] (AF32)
    float* ssq_s = wst + (AF32 ? NW * 256 : 0);                                          // [NW][16]
    float* amax_s = ssq_s + NW * 16;                                                     // [NFRAG][16]
    int* aidx_s = reinterpret_cast<int*>(amax_s + NFRAG * 16);                           // [NFRAG][16]

    // the residual quad of the fragment this wave finishes (wave f finishes fragment f): requested first, needed last
    float4 rpre = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const int n0 = n_tile + w * 16 + kq * 4;
        if (a.resid && w < NFRAG && m < a.rows && n0 < a.N) rpre = *reinterpret_cast<const float4*>(a.resid + (size_t)m * a.ldr + n0);
    }
    // producer of a split norm: the next norm's weights of the same quad; consumer: this lane's share of the row's partial sums of squares
    float4 wpre = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 rsp[4] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    {
        const int n0 = n_tile + w * 16 + kq * 4;
        if (a.nx_b && w < NFRAG && n0 < a.N) wpre = *reinterpret_cast<const float4*>(a.nx_w + n0);
        if constexpr (!AF32) {
            if (a.rs_part && w < NFRAG) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i * 16 + kq * 4 < a.rs_n) rsp[i] = *reinterpret_cast<const float4*>(a.rs_part + (size_t)mc * a.rs_n + i * 16 + kq * 4);
            }
        }
    }
    const int t_step = a.t_ptr ? *a.t_ptr : 0;
    f32x4_t acc[NFRAG];
#pragma unroll
    for (int f = 0; f < NFRAG; ++f) acc[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float ssq = 0.f;
    const int prow = lane >> 3, pchunk = (lane & 7) ^ (lane >> 3);           // this lane's (row within a piece, source chunk)
    const int phases = a.K / (64 * KT * NW);
    for (int ph = 0; ph < phases; ++ph) {
        const int k0 = (ph * NW + w) * (KT * 64);                            // this wave's k-range of the phase: KT tiles of 64
        // ---- requests of the phase, in the order they are consumed: norm weights + f32 rows (registers), then the DMA pieces ----
        float4 xa[AF32 ? KT * 2 : 1][2];
        float4 wl = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (AF32) {
            if (lane * 4 < KT * 64) wl = *reinterpret_cast<const float4*>(a.ln_w + k0 + lane * 4);
            const float* xp = a.xf + (size_t)mc * a.ldx + k0 + kq * 8;
#pragma unroll
            for (int ks = 0; ks < KT * 2; ++ks) {
                xa[ks][0] = *reinterpret_cast<const float4*>(xp + ks * 32);
                xa[ks][1] = *reinterpret_cast<const float4*>(xp + ks * 32 + 4);
            }
        }
        if (ph > 0) {                                                        // the stage is reused: every read of the phase before is done
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            if constexpr (!AF32) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    int mr = rb * 16 + i * 8 + prow;
                    mr = mr < a.rows ? mr : a.rows - 1;
                    __builtin_amdgcn_global_load_lds(a.xb + (size_t)mr * a.ldx + k0 + kt * 64 + pchunk * 8,
                                                     (dl_lds_ptr_t)(stage + KT * WT_BYTES + kt * XT_BYTES + i * 1024), 16, 0, DECLIN_X_AUX);
                }
            }
#pragma unroll
            for (int i = 0; i < NFRAG * 2; ++i) {
                int n = n_tile + i * 8 + prow;
                n = n < a.N ? n : a.N - 1;
                __builtin_amdgcn_global_load_lds(a.W + (size_t)n * a.K + k0 + kt * 64 + pchunk * 8,
                                                 (dl_lds_ptr_t)(stage + kt * WT_BYTES + i * 1024), 16, 0, DECLIN_W_AUX);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ph == 0) TLS(1);
        // ---- f32 activation rows -> bf16(x * w_norm) fragments + the rows' sums of squares (while the weight pieces are on their way) ----
        bf16x8_t fx[KT * 2];
        if constexpr (AF32) {
            float* strip = wst + w * 256;
            if (lane * 4 < KT * 64) *reinterpret_cast<float4*>(strip + lane * 4) = wl;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (the strip is this wave's own: no barrier)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < KT * 2; ++ks) {
                const float4 w0 = *reinterpret_cast<const float4*>(strip + ks * 32 + kq * 8);
                const float4 w1 = *reinterpret_cast<const float4*>(strip + ks * 32 + kq * 8 + 4);
                const float4 x0 = xa[ks][0], x1 = xa[ks][1];
                ssq += x0.x * x0.x + x0.y * x0.y + x0.z * x0.z + x0.w * x0.w + x1.x * x1.x + x1.y * x1.y + x1.z * x1.z + x1.w * x1.w;
                const uint4 pk = make_uint4(pack_bf16x2(x0.x * w0.x, x0.y * w0.y), pack_bf16x2(x0.z * w0.z, x0.w * w0.w),
                                            pack_bf16x2(x1.x * w1.x, x1.y * w1.y), pack_bf16x2(x1.z * w1.z, x1.w * w1.w));
                fx[ks] = __builtin_bit_cast(bf16x8_t, pk);
            }
            if (ph == 0) TLS(2);
        }
        // ---- the wave's own pieces have landed: fragments out of LDS, MFMAs ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                const int slot = (p2 * 4 + kq) ^ (r16 & 7);
                bf16x8_t fa;
                if constexpr (AF32) fa = fx[kt * 2 + p2];
                else fa = *reinterpret_cast<const bf16x8_t*>(stage + KT * WT_BYTES + kt * XT_BYTES + r16 * 128 + slot * 16);
#pragma unroll
                for (int f = 0; f < NFRAG; ++f) {          // weight fragment as the A operand: lane (r16, kq) ends up with out[m = r16][n = kq*4 .. +3]
                    const bf16x8_t fw = *reinterpret_cast<const bf16x8_t*>(stage + kt * WT_BYTES + (f * 16 + r16) * 128 + slot * 16);
                    acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, fa, acc[f], 0, 0, 0);
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    TLS(3);
    // ---- partial tiles of the NW waves meet in LDS ----------------------------------------------------------------------
#pragma unroll
    for (int f = 0; f < NFRAG; ++f) red[(w * NFRAG + f) * 64 + lane] = make_float4(acc[f][0], acc[f][1], acc[f][2], acc[f][3]);
    if constexpr (AF32) {
        ssq += __shfl_xor(ssq, 16, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (lane < 16) ssq_s[w * 16 + lane] = ssq;
    }
    __syncthreads();
    TLS(4);
    float rs = a.alpha;
    if constexpr (AF32) {
        float s = 0.f;
        for (int ww = 0; ww < NW; ++ww) s += ssq_s[ww * 16 + r16];
        rs *= rsqrtf(s / (float)a.K + a.eps);
    } else if (a.rs_part) {                                       // (waves >= NFRAG hold zeros and finish no fragment)
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += (rsp[i].x + rsp[i].y) + (rsp[i].z + rsp[i].w);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        rs *= rsqrtf(s / (float)a.K + a.eps);
    }
    for (int f = w; f < NFRAG; f += NW) {
        float4 v = red[f * 64 + lane];
        for (int ww = 1; ww < NW; ++ww) {
            const float4 o = red[(ww * NFRAG + f) * 64 + lane];
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs;
        const int n0 = n_tile + f * 16 + kq * 4;
        const bool ok = m < a.rows && n0 < a.N;                  // (N is a multiple of 4: a quad is inside or outside as a whole)
        if (a.resid && ok) {
            float4 r = rpre;
            if (f != w) r = *reinterpret_cast<const float4*>(a.resid + (size_t)m * a.ldr + n0);     // (fewer waves than fragments: tiny shapes)
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (a.nx_b) {                                              // (workgroup-uniform) split norm: operand of the next projection + this fragment's sum of squares
            float q = ok ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : 0.f;
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            if (lane < 16 && m < a.rows && n_tile + f * 16 < a.N) a.nx_ssq[(size_t)m * a.nx_parts + (n_tile >> 4) + f] = q;
            if (ok) {
                float4 wq = wpre;
                if (f != w) wq = *reinterpret_cast<const float4*>(a.nx_w + n0);
                dl_store8(a.nx_b + (size_t)m * a.ld_nx + n0, make_uint2(pack_bf16x2(v.x * wq.x, v.y * wq.y), pack_bf16x2(v.z * wq.z, v.w * wq.w)));
            }
        }
        if (ok) {
            if (a.out_f) dl_store16f(a.out_f + (size_t)m * a.ldf + n0, v);
            if (a.out_b) {
                const uint2 pk = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                if (n0 < a.split_col) dl_store8(a.out_b + (size_t)m * a.ldo + n0, pk);
                else dl_store8(a.out_b2 + (size_t)t_step * a.t_stride2 + (size_t)m * a.ldo2 + (n0 - a.split_col), pk);
            }
        }
        if (a.pmax) {
            // first maximum of the row inside this 16-column fragment (lowest column wins a tie)
            float best = -INFINITY;
            int bi = 0x7fffffff;
            if (ok) {
                best = v.x; bi = n0;
                if (v.y > best) { best = v.y; bi = n0 + 1; }
                if (v.z > best) { best = v.z; bi = n0 + 2; }
                if (v.w > best) { best = v.w; bi = n0 + 3; }
            }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float ov = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (lane < 16) { amax_s[f * 16 + lane] = best; aidx_s[f * 16 + lane] = bi; }
        }
    }
    if (a.pmax) {
        __syncthreads();
        if (threadIdx.x < 16 && rb * 16 + (int)threadIdx.x < a.rows) {
            float best = amax_s[threadIdx.x];
            int bi = aidx_s[threadIdx.x];
#pragma unroll
            for (int f = 1; f < NFRAG; ++f) {
                const float ov = amax_s[f * 16 + threadIdx.x];
                const int oi = aidx_s[f * 16 + threadIdx.x];
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            const size_t slot = (size_t)(rb * 16 + threadIdx.x) * a.CT + ct;
            a.pmax[slot] = best;
            a.pidx[slot] = bi;
        }
    }
    TLS(5);
#ifdef DECLIN_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TLS(6);
#endif
    if (a.t_inc && blockIdx.x == 0 && threadIdx.x == 0) *a.t_inc += 1;          // (no workgroup of this launch reads the index: t_ptr is null)
}

// The vocabulary projection: N = 32200 columns make 504 column tiles -- enough workgroups without cutting the rows, and its 49.5 MB weight
// is the one operand of the step that is worth reading ONCE: a workgroup stages its 64-column slice a single time and walks the row blocks
// (the f32 rows of block rb + 1 are requested while block rb is computed).  Norm folded in as in declin_kernel<true, ...>; one phase only.
// The reduction is split over the waves as above, but the waves' partial tiles meet ONCE per chunk of RBC row blocks instead of once per
// row block: a wave keeps the accumulators of the whole chunk (RBC x NFRAG tiles), after the last MFMA the partials overwrite the weight
// stage (nobody reads it any more) and wave w finishes row blocks w, w + NW, ...: sums, rstd, the logits store if asked for, and the
// (row, column tile) argmax partial, all four fragments in its own registers.  Three barriers per chunk (B = 80: one chunk) against two
// per row block before -- the walk was a chain of five reduce / barrier / epilogue rounds per workgroup, 39 us per step.
template <bool AF32, int KT, int NFRAG>
__global__ __launch_bounds__(256) void declin_rows_kernel(const DecLinArgs a) {      // (<= 4 waves: the chunk's accumulators + two sets of f32 rows need > 256 registers)
    constexpr int RBC = KT * 2 < 5 ? KT * 2 : 5;                   // row blocks per chunk: their partial tiles fit the weight stage
    extern __shared__ __attribute__((aligned(16))) char dl_smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const int ct = blockIdx.x;
    const int r16 = lane & 15, kq = lane >> 4;
    const int n_tile = ct * (16 * NFRAG);
    constexpr int WT_BYTES = NFRAG * 16 * 128, WAVE_BYTES = KT * WT_BYTES;
    static_assert(RBC * NFRAG * 1024 <= WAVE_BYTES, "a wave's partial tiles of a chunk fit its share of the stage");
    char* stage = dl_smem + (size_t)w * WAVE_BYTES;
    float4* red = reinterpret_cast<float4*>(dl_smem);                                     // [NW][RBC][NFRAG][64], over the stage
    float* wst = reinterpret_cast<float*>(dl_smem + (size_t)NW * WAVE_BYTES);             // [NW][256]
    float* ssq_s = wst + NW * 256;                                                        // [RBC][NW][16]
    const int prow = lane >> 3, pchunk = (lane & 7) ^ (lane >> 3);
    const int k0 = w * (KT * 64);
    constexpr int XV = AF32 ? 2 : 1;                               // 16-byte loads per lane and 32-deep k-slab: 8 f32 or 8 bf16
    float4 wl = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (AF32) { if (lane * 4 < KT * 64) wl = *reinterpret_cast<const float4*>(a.ln_w + k0 + lane * 4); }
    // split norm (bf16 rows emitted by the producer of the residual stream): wave w finishes row blocks w and w + NW of a chunk -- their
    // partial sums of squares are requested first
    float4 rsp[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) rsp[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_parts = [&](int rb0) {
        if constexpr (!AF32) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = w + u * NW;
                int mr = (rb0 + c) * 16 + r16;
                mr = mr < a.rows ? mr : a.rows - 1;
                if (c < RBC && rb0 + c < a.RB) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i * 16 + kq * 4 < a.rs_n) rsp[u][i] = *reinterpret_cast<const float4*>(a.rs_part + (size_t)mr * a.rs_n + i * 16 + kq * 4);
                }
            }
        }
    };
    auto load_rows = [&](int rb, float4 (&x)[KT * 2][XV]) {
        int mr = rb * 16 + r16;
        mr = mr < a.rows ? mr : a.rows - 1;
        if constexpr (AF32) {
            const float* xp = a.xf + (size_t)mr * a.ldx + k0 + kq * 8;
#pragma unroll
            for (int ks = 0; ks < KT * 2; ++ks) {
                x[ks][0] = *reinterpret_cast<const float4*>(xp + ks * 32);
                x[ks][XV - 1] = *reinterpret_cast<const float4*>(xp + ks * 32 + 4);
            }
        } else {
            const bf16_t* xp = a.xb + (size_t)mr * a.ldx + k0 + kq * 8;
#pragma unroll
            for (int ks = 0; ks < KT * 2; ++ks) x[ks][0] = *reinterpret_cast<const float4*>(xp + ks * 32);
        }
    };
    auto stage_w = [&]() {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int i = 0; i < NFRAG * 2; ++i) {
                int n = n_tile + i * 8 + prow;
                n = n < a.N ? n : a.N - 1;
                __builtin_amdgcn_global_load_lds(a.W + (size_t)n * a.K + k0 + kt * 64 + pchunk * 8, (dl_lds_ptr_t)(stage + kt * WT_BYTES + i * 1024), 16, 0, DECLIN_ROWS_W_AUX);
            }
        }
    };
    float4 xa[KT * 2][XV], xn[KT * 2][XV];
    load_parts(0);
    load_rows(0, xa);
    stage_w();
    float* strip = wst + w * 256;
    if constexpr (AF32) { if (lane * 4 < KT * 64) *reinterpret_cast<float4*>(strip + lane * 4) = wl; }
    for (int rb0 = 0; rb0 < a.RB; rb0 += RBC) {
        if (rb0 > 0) {                                             // (more than RBC row blocks: the stage held partial tiles -- fetch the slice again)
            __syncthreads();
            load_parts(rb0);
            load_rows(rb0, xa);
            stage_w();
        }
        f32x4_t acc[RBC][NFRAG];
#pragma unroll
        for (int c = 0; c < RBC; ++c)
#pragma unroll
            for (int f = 0; f < NFRAG; ++f) acc[c][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < RBC; ++c) {
            const int rb = rb0 + c;
            if (rb < a.RB) {                                       // (workgroup-uniform)
                if (c + 1 < RBC && rb + 1 < a.RB) load_rows(rb + 1, xn);
                __builtin_amdgcn_sched_barrier(0);
                bf16x8_t fx[KT * 2];
                if constexpr (AF32) {
                    float ssq = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KT * 2; ++ks) {
                        const float4 w0 = *reinterpret_cast<const float4*>(strip + ks * 32 + kq * 8);
                        const float4 w1 = *reinterpret_cast<const float4*>(strip + ks * 32 + kq * 8 + 4);
                        const float4 x0 = xa[ks][0], x1 = xa[ks][XV - 1];
                        ssq += x0.x * x0.x + x0.y * x0.y + x0.z * x0.z + x0.w * x0.w + x1.x * x1.x + x1.y * x1.y + x1.z * x1.z + x1.w * x1.w;
                        const uint4 pk = make_uint4(pack_bf16x2(x0.x * w0.x, x0.y * w0.y), pack_bf16x2(x0.z * w0.z, x0.w * w0.w),
                                                    pack_bf16x2(x1.x * w1.x, x1.y * w1.y), pack_bf16x2(x1.z * w1.z, x1.w * w1.w));
                        fx[ks] = __builtin_bit_cast(bf16x8_t, pk);
                    }
                    ssq += __shfl_xor(ssq, 16, 64);
                    ssq += __shfl_xor(ssq, 32, 64);
                    if (lane < 16) ssq_s[(c * NW + w) * 16 + lane] = ssq;
                } else {
#pragma unroll
                    for (int ks = 0; ks < KT * 2; ++ks) fx[ks] = __builtin_bit_cast(bf16x8_t, xa[ks][0]);
                }
                if (c == 0) {                                      // the wave's weight pieces have landed (the rows of block rb + 1 stay in flight)
                    if (RBC > 1 && rb + 1 < a.RB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KT * 2 * XV) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                    for (int p2 = 0; p2 < 2; ++p2) {
                        const int slot = (p2 * 4 + kq) ^ (r16 & 7);
#pragma unroll
                        for (int f = 0; f < NFRAG; ++f) {
                            const bf16x8_t fw = *reinterpret_cast<const bf16x8_t*>(stage + kt * WT_BYTES + (f * 16 + r16) * 128 + slot * 16);
                            acc[c][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, fx[kt * 2 + p2], acc[c][f], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int ks = 0; ks < KT * 2; ++ks) { xa[ks][0] = xn[ks][0]; xa[ks][XV - 1] = xn[ks][XV - 1]; }
            }
        }
        __syncthreads();                                           // every wave is done with the weight stage
#pragma unroll
        for (int c = 0; c < RBC; ++c)
#pragma unroll
            for (int f = 0; f < NFRAG; ++f)
                red[((w * RBC + c) * NFRAG + f) * 64 + lane] = make_float4(acc[c][f][0], acc[c][f][1], acc[c][f][2], acc[c][f][3]);
        __syncthreads();
        for (int c = w; c < RBC; c += NW) {
            const int rb = rb0 + c;
            if (rb >= a.RB) break;
            float s = 0.f;
            if constexpr (AF32) {
                for (int ww = 0; ww < NW; ++ww) s += ssq_s[(c * NW + ww) * 16 + r16];
            } else {
                const int u = c >= NW ? 1 : 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) s += (u ? (rsp[1][i].x + rsp[1][i].y) + (rsp[1][i].z + rsp[1][i].w) : (rsp[0][i].x + rsp[0][i].y) + (rsp[0][i].z + rsp[0][i].w));
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
            }
            const float rs = a.alpha * rsqrtf(s / (float)a.K + a.eps);
            const int m = rb * 16 + r16;
            float best = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int f = 0; f < NFRAG; ++f) {
                float4 v = red[(c * NFRAG + f) * 64 + lane];
                for (int ww = 1; ww < NW; ++ww) {
                    const float4 o = red[((ww * RBC + c) * NFRAG + f) * 64 + lane];
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs;
                const int n0 = n_tile + f * 16 + kq * 4;
                const bool ok = m < a.rows && n0 < a.N;
                if (ok && a.out_f) dl_store16f(a.out_f + (size_t)m * a.ldf + n0, v);
                if (ok) {                                          // first maximum wins: columns ascend with f, then within the quad
                    if (v.x > best) { best = v.x; bi = n0; }
                    if (v.y > best) { best = v.y; bi = n0 + 1; }
                    if (v.z > best) { best = v.z; bi = n0 + 2; }
                    if (v.w > best) { best = v.w; bi = n0 + 3; }
                }
            }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {                   // the four lanes of a row (their quads interleave: ties go to the lower column)
                const float ov = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (a.pmax && lane < 16 && m < a.rows) {
                const size_t slot = (size_t)m * a.CT + ct;
                a.pmax[slot] = best;
                a.pidx[slot] = bi;
            }
        }
    }
    if (a.t_inc && blockIdx.x == 0 && threadIdx.x == 0) *a.t_inc += 1;
}

// The vocabulary projection of the decode step, second form (split norm: bf16 rows + partial sums of squares from the producer; <= 8 row
// blocks).  What bounded declin_rows_kernel: 504 workgroups = two rounds per CU, each re-reading all rows (123 KB bf16 beside its 98 KB
// weight slice) and with nothing in flight while it reduces.  Here a workgroup is RESIDENT (<= one per CU) and walks 32-column tiles:
//   * wave w owns row block w for the whole launch: its 16 rows x K as MFMA fragments stay in registers (K / 32 x 4 VGPRs), requested once;
//   * a tile's weight slice (32 x K bf16 = 48 KB at K = 768) is staged by all waves together, double-buffered: the pieces of tile i + 1 are
//     requested before tile i is computed, so the 49.5 MB weight streams behind the MFMAs instead of in front of them;
//   * no reduction split over waves: a wave reads every weight fragment of the tile (each feeds its one row block), finishes its rows
//     itself -- rstd from the partial sums, the logits store if asked for, the first maximum of the row inside the tile -- and ONE
//     workgroup barrier per tile orders the stage hand-over (everybody's pieces of tile i landed / everybody done with tile i - 1).
// Tiles per row: ceil(N / 32) (vlt5_declin_vocab_tiles); tile t of workgroup b: b + t * gridDim.
template <int KTT>                                              // K / 64
__global__ __launch_bounds__(512) void declin_vocab_kernel(const DecLinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dl_smem[];
    constexpr int TILE_ROWS = 32, PIECES = TILE_ROWS * KTT / 8;   // 1 KB pieces (8 weight rows x 128 B) of a tile: [kt][row / 8]
    constexpr int STAGE_BYTES = PIECES * 1024;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int prow = lane >> 3, pchunk = (lane & 7) ^ (lane >> 3);
    const int ppw = (PIECES + NW - 1) / NW;                       // pieces per wave (the last ones repeat piece PIECES - 1: a uniform vmcnt)
    const int ntiles = (a.N + TILE_ROWS - 1) / TILE_ROWS;
    int mr = w * 16 + r16;
    const int m = mr;
    mr = mr < a.rows ? mr : a.rows - 1;
    // this lane's share of the row's partial sums of squares, then the row itself as fragments
    float4 rsp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rsp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i * 16 + kq * 4 < a.rs_n) rsp[i] = *reinterpret_cast<const float4*>(a.rs_part + (size_t)mr * a.rs_n + i * 16 + kq * 4);
    }
    bf16x8_t fx[KTT * 2];
    {
        const bf16_t* xp = a.xb + (size_t)mr * a.ldx + kq * 8;
#pragma unroll
        for (int ks = 0; ks < KTT * 2; ++ks) fx[ks] = *reinterpret_cast<const bf16x8_t*>(xp + ks * 32);
    }
    auto stage_tile = [&](int tile, int buf) {                   // this wave's pieces of the tile -> stage `buf`
        const int n_tile = tile * TILE_ROWS;
        char* st = dl_smem + (size_t)buf * STAGE_BYTES;
        for (int i = 0; i < ppw; ++i) {
            int pc = w + i * NW;
            pc = pc < PIECES ? pc : PIECES - 1;
            const int kt = pc / (TILE_ROWS / 8), rg = pc % (TILE_ROWS / 8);
            int n = n_tile + rg * 8 + prow;
            n = n < a.N ? n : a.N - 1;
            __builtin_amdgcn_global_load_lds(a.W + (size_t)n * a.K + kt * 64 + pchunk * 8, (dl_lds_ptr_t)(st + (size_t)pc * 1024), 16, 0, DECLIN_ROWS_W_AUX);
        }
    };
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (rsp[i].x + rsp[i].y) + (rsp[i].z + rsp[i].w);
    int tile = blockIdx.x, it = 0;
    if (tile < ntiles) stage_tile(tile, 0);
    s += __shfl_xor(s, 16, 64);                                    // (the first use of the partials: behind the first tile's requests)
    s += __shfl_xor(s, 32, 64);
    const float rs = a.alpha * rsqrtf(s / (float)a.K + a.eps);
    for (; tile < ntiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // my pieces of this tile (and, first time, my rows) have landed
        __syncthreads();                                         // everybody's have; everybody is done with the other stage
        if (tile + (int)gridDim.x < ntiles) stage_tile(tile + gridDim.x, buf ^ 1);
        const char* st = dl_smem + (size_t)buf * STAGE_BYTES;
        f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kt = 0; kt < KTT; ++kt) {
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                const int slot = (p2 * 4 + kq) ^ (r16 & 7);
#pragma unroll
                for (int f = 0; f < 2; ++f) {                     // piece (kt, rg) holds weight rows rg*8 .. +7 of the tile: row f*16 + r16
                    const bf16x8_t fw = *reinterpret_cast<const bf16x8_t*>(st + (size_t)(kt * (TILE_ROWS / 8)) * 1024 + (f * 16 + r16) * 128 + slot * 16);
                    acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, fx[kt * 2 + p2], acc[f], 0, 0, 0);
                }
            }
        }
        const int n_tile = tile * TILE_ROWS;
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const float4 v = make_float4(acc[f][0] * rs, acc[f][1] * rs, acc[f][2] * rs, acc[f][3] * rs);
            const int n0 = n_tile + f * 16 + kq * 4;
            const bool ok = m < a.rows && n0 < a.N;
            if (ok && a.out_f) dl_store16f(a.out_f + (size_t)m * a.ldf + n0, v);
            if (ok) {                                              // first maximum wins: columns ascend with f, then within the quad
                if (v.x > best) { best = v.x; bi = n0; }
                if (v.y > best) { best = v.y; bi = n0 + 1; }
                if (v.z > best) { best = v.z; bi = n0 + 2; }
                if (v.w > best) { best = v.w; bi = n0 + 3; }
            }
        }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (a.pmax && lane < 16 && m < a.rows) {
            const size_t slot = (size_t)m * ntiles + tile;
            a.pmax[slot] = best;
            a.pidx[slot] = bi;
        }
    }
    if (a.t_inc && blockIdx.x == 0 && threadIdx.x == 0) *a.t_inc += 1;
}

typedef void (*declin_fn)(const DecLinArgs);
template <bool AF32, int KT>
declin_fn declin_pick_nfrag(int nfrag) {
    switch (nfrag) {
        case 1: return &declin_kernel<AF32, KT, 1>;
        case 2: return &declin_kernel<AF32, KT, 2>;
        case 3: return &declin_kernel<AF32, KT, 3>;
        case 4: return &declin_kernel<AF32, KT, 4>;
        default: return nullptr;
    }
}
template <bool AF32>
declin_fn declin_pick(int kt, int nfrag) {
    switch (kt) {
        case 1: return declin_pick_nfrag<AF32, 1>(nfrag);
        case 2: return declin_pick_nfrag<AF32, 2>(nfrag);
        case 3: return declin_pick_nfrag<AF32, 3>(nfrag);
        case 4: return declin_pick_nfrag<AF32, 4>(nfrag);
        default: return nullptr;
    }
}
// dynamic LDS of a launch: per-wave stages + partial tiles + norm-weight strips + sums of squares + argmax exchange
static size_t declin_lds_bytes(bool af32, int kt, int nfrag, int nw) {
    const size_t wave = (size_t)kt * (nfrag * 16 * 128 + (af32 ? 0 : 16 * 128));
    return nw * wave + (size_t)nw * nfrag * 64 * 16 + (af32 ? nw * 256 * 4 : 0) + nw * 16 * 4 + nfrag * 16 * 8;
}

// ------------------------------------------------------------------------------------------------------------------------------
// attention core of one new token: one wave per (sample, head)
// ------------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned dl_u32x4_t;
__device__ __forceinline__ uint4 core_load16(const bf16_t* p) {
#if DEC_CORE_NT
    const dl_u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const dl_u32x4_t*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
#else
    return *reinterpret_cast<const uint4*>(p);
#endif
}
template <int CH>        // CH = d_kv / 16 lanes share a key (16 dims each); 64 / CH keys per pass, CH passes cover 64 keys
__global__ __launch_bounds__(256) void dec_core_kernel(const DecCoreArgs a) {
    constexpr int KSL = 64 / CH, LOGC = CH == 4 ? 2 : (CH == 2 ? 1 : 0);
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= a.B * a.H) return;
    const int b = gw / a.H, h = gw % a.H;
    const int c = lane & (CH - 1), slot = lane >> LOGC;
    const int Tk = a.t_ptr ? *a.t_ptr + 1 : a.Tk;
    const int dk = CH * 16;
    const bf16_t* qp = a.q + (size_t)b * a.q_ld + h * dk + c * 16;
    const uint4 q0 = *reinterpret_cast<const uint4*>(qp), q1 = *reinterpret_cast<const uint4*>(qp + 8);
    uint4 kk[CH][2], vv[CH][2];
    float add[CH];
    const bf16_t* kb = a.k + (size_t)b * a.kv_sb + h * dk + c * 16;
    const bf16_t* vb = a.v + (size_t)b * a.kv_sb + h * dk + c * 16;
#pragma unroll
    for (int p = 0; p < CH; ++p) {
        const int key = p * KSL + slot;
        const int kc = key < Tk ? key : 0;
        kk[p][0] = core_load16(kb + (size_t)kc * a.kv_st);
        kk[p][1] = core_load16(kb + (size_t)kc * a.kv_st + 8);
        vv[p][0] = core_load16(vb + (size_t)kc * a.kv_st);
        vv[p][1] = core_load16(vb + (size_t)kc * a.kv_st + 8);
        float ad = 0.f;
        if (a.bias) ad += a.bias[(size_t)h * a.bias_ld + kc];                                  // relative-position bias row of the step
        if (a.key_mask) ad += (1.0f - a.key_mask[(size_t)b * a.mask_ld + kc]) * a.mask_value;    // padded encoder-side keys
        add[p] = ad;
    }
    auto lo = [](uint32_t u) { return __uint_as_float(u << 16); };
    auto hi = [](uint32_t u) { return __uint_as_float(u & 0xffff0000u); };
    auto dot8 = [&](const uint4& x, const uint4& y) {
        return lo(x.x) * lo(y.x) + hi(x.x) * hi(y.x) + lo(x.y) * lo(y.y) + hi(x.y) * hi(y.y) + lo(x.z) * lo(y.z) + hi(x.z) * hi(y.z) +
               lo(x.w) * lo(y.w) + hi(x.w) * hi(y.w);
    };
    float s[CH], mx = -INFINITY;
#pragma unroll
    for (int p = 0; p < CH; ++p) {
        float d = dot8(q0, kk[p][0]) + dot8(q1, kk[p][1]);
#pragma unroll
        for (int o = 1; o < CH; o <<= 1) d += __shfl_xor(d, o, 64);      // the CH lanes of a key
        s[p] = (p * KSL + slot < Tk) ? d + add[p] : -INFINITY;
        mx = fmaxf(mx, s[p]);
    }
#pragma unroll
    for (int o = CH; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f, acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int p = 0; p < CH; ++p) {
        const float e = (p * KSL + slot < Tk) ? fast_exp(s[p] - mx) : 0.f;
        sum += e;
        const uint32_t vw[8] = {vv[p][0].x, vv[p][0].y, vv[p][0].z, vv[p][0].w, vv[p][1].x, vv[p][1].y, vv[p][1].z, vv[p][1].w};
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[2 * i] += e * lo(vw[i]); acc[2 * i + 1] += e * hi(vw[i]); }
    }
#pragma unroll
    for (int o = CH; o < 64; o <<= 1) sum += __shfl_xor(sum, o, 64);
    // sum over the key slots (lane bits LOGC..5): halving exchange on lane bits 5, 4, 3, 2 (16 -> 8 -> 4 -> 2 -> 1 values per lane),
    // plain sums on the key-slot bits below 2; lane ends up with dimension c*16 + ((lane >> 2) & 15)
    {
        float v8[8];
        const bool up5 = lane & 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float snd = up5 ? acc[i] : acc[i + 8]; const float kp = up5 ? acc[i + 8] : acc[i]; v8[i] = kp + __shfl_xor(snd, 32, 64); }
        float v4[4];
        const bool up4 = lane & 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float snd = up4 ? v8[i] : v8[i + 4]; const float kp = up4 ? v8[i + 4] : v8[i]; v4[i] = kp + __shfl_xor(snd, 16, 64); }
        float v2[2];
        const bool up3 = lane & 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) { const float snd = up3 ? v4[i] : v4[i + 2]; const float kp = up3 ? v4[i + 2] : v4[i]; v2[i] = kp + __shfl_xor(snd, 8, 64); }
        const bool up2 = lane & 4;
        float r = (up2 ? v2[1] : v2[0]) + __shfl_xor(up2 ? v2[0] : v2[1], 4, 64);
#pragma unroll
        for (int o = CH; o < 4; o <<= 1) r += __shfl_xor(r, o, 64);      // (d_kv < 64: lane bits 0/1 are key-slot bits too)
        const int dim = c * 16 + ((lane >> 2) & 15);
        const bool writer = CH == 4 || (CH == 2 && !(lane & 2)) || (CH == 1 && !(lane & 3));
        if (writer) a.ctx[(size_t)b * a.ctx_ld + h * dk + dim] = f32_to_bf16(r / sum);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// between two steps: argmax finish + greedy bookkeeping + next input row + next bias row.  One workgroup per sample.
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_io_kernel(const DecIoArgs a) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    __shared__ long long tok_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int tq = a.t_ptr ? *a.t_ptr : a.tq, out_col = a.t_ptr ? tq : a.out_col;        // (device-side step index: see vlt5_greedy_desc.t_dev)
    const bool emit_next = !a.t_ptr || tq < a.Tcap;
    long long tok;
    if (a.pmax) {
        float best = -INFINITY;
        int idx = 0x7fffffff;
        for (int c = tid; c < a.ptiles; c += 256) {
            const float v = a.pmax[(size_t)b * a.ptiles + c];
            const int i = a.pidx[(size_t)b * a.ptiles + c];
            if (v > best || (v == best && i < idx)) { best = v; idx = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(idx, o, 64);
            if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
        }
        if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int w = 1; w < 4; ++w)
                if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
            long long nxt = idx == 0x7fffffff ? 0 : idx;          // an all-NaN row: index 0 (as vlt5_argmax_rows)
            if (a.next_ids) a.next_ids[b] = nxt;
            if (a.done) {                                          // HF greedy search: a finished row keeps emitting pad
                const int was = a.done[b];
                if (was) nxt = a.pad_id;
                a.done[b] = was | (nxt == a.eos_id);
                if (a.out_tokens) a.out_tokens[(size_t)b * a.out_ld + out_col] = nxt;
            }
            tok_s = nxt;
        }
        __syncthreads();
        tok = tok_s;
    } else {
        tok = a.tokens[b];
    }
    if (a.emb_out && emit_next) {                                  // next step's decoder input row (embedding lookup, eval: no dropout)
        long long id = tok < 0 ? 0 : (tok >= a.vocab ? a.vocab - 1 : tok);
        const float4* src = reinterpret_cast<const float4*>(a.table + (size_t)id * a.d);
        float4* dst = reinterpret_cast<float4*>(a.emb_out + (size_t)b * a.d);
        float q = 0.f;
        for (int i = tid; i < a.d / 4; i += 256) {
            const float4 v = src[i];
            dst[i] = v;
            if (a.nx_b) {                                          // split norm (DecLinArgs.nx_*): the operand of the first projection of layer 0
                const float4 wq = reinterpret_cast<const float4*>(a.nx_w)[i];
                q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                *reinterpret_cast<uint2*>(a.nx_b + (size_t)b * a.d + i * 4) = make_uint2(pack_bf16x2(v.x * wq.x, v.y * wq.y), pack_bf16x2(v.z * wq.z, v.w * wq.w));
            }
        }
        if (a.nx_b) {
            __shared__ float qs[4];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
            if ((tid & 63) == 0) qs[tid >> 6] = q;
            __syncthreads();
            for (int j = tid; j < a.nx_parts; j += 256) a.nx_ssq[(size_t)b * a.nx_parts + j] = j == 0 ? (qs[0] + qs[1]) + (qs[2] + qs[3]) : 0.f;
        }
    }
    if (a.bias_out && b == 0 && emit_next) {                       // bias row of query position tq against keys 0..tq: [H][bias_ld]
        const int n = a.H * (tq + 1);
        for (int i = tid; i < n; i += 256) {
            const int h = i / (tq + 1), jk = i % (tq + 1);
            a.bias_out[(size_t)h * a.bias_ld + jk] = a.rel_table[(size_t)a.lut[(size_t)tq * a.lut_ld + jk] * a.H + h];
        }
    }
}

}  // namespace

// K = 64 * KT * NW * phases: k-tiles of 64 per wave and phase (KT <= 4, the per-wave stages must fit the LDS), NW <= 8 waves; fewest
// phases first (each is a memory round trip), then four waves before eight
static bool declin_split(int K, bool af32, int nfrag, int* kt, int* nw) {
    if (K <= 0 || (K & 63)) return false;
    const int tiles = K / 64;
    int best_ph = 1 << 30;
    const int tries[4] = {4, 8, 2, 1};
    for (int t = 0; t < 4; ++t) {
        const int w = tries[t];
        if (tiles % w) continue;
        const int per = tiles / w;
        for (int k = 4; k >= 1; --k) {
            if (per % k || declin_lds_bytes(af32, k, nfrag, w) > 160 * 1024) continue;
            if (per / k < best_ph) { best_ph = per / k; *kt = k; *nw = w; }
            break;
        }
    }
    return best_ph < (1 << 30);
}

// the launch geometry of (rows, N, K): k-tiles per wave and phase, waves, fragments per column tile
static bool declin_geometry(int rows, int N, int K, bool af32, int* kt, int* nw, int* nfrag, int rule = 0) {
    const int RB = (rows + 15) / 16;
    // More than five row blocks (> 80 rows; the reference evaluates with --valid_batch_size 100): a wide tile then means MORE than one round
    // of heavy workgroups, and narrow tiles win -- two fragments if that makes the grid one round of the chip, else one (round 5, rows 96 ... 320:
    // -8 % per batch at 100 rows, -13 % at 320; profiles/r05_y_decode_rules.txt).  The vocabulary projection keeps its own forms.
    if (RB >= 6 && N < 8192 && rule != 16) {
        const int f = ((N + 31) / 32) * RB <= 256 ? 2 : 1;
        *nfrag = f;
        return declin_split(K, af32, f, kt, nw);
    }
    for (int f = 4; f >= 1; --f) {          // widest column tile that still fills the chip (>= ~200 workgroups)
        const int ctf = (N + 16 * f - 1) / (16 * f);
        if (ctf * RB >= 200 || f == 1) { *nfrag = f; return declin_split(K, af32, f, kt, nw); }
    }
    return false;
}
extern "C" int vlt5_decode_linear_supported(int K, int af32) {
    int kt, nw;
    return declin_split(K, af32 != 0, 4, &kt, &nw) ? 1 : 0;       // (true for the widest column tile: true for every narrower one)
}

#ifdef DECLIN_TIMELINE
static long long* g_declin_tl = nullptr;       // instrumented builds only (never the product library): next stamp block
static long long g_declin_tl_stride = 0;
extern "C" void vlt5_declin_timeline(long long* buf, long long stride_per_launch) { g_declin_tl = buf; g_declin_tl_stride = stride_per_launch; }
#endif
// the resident vocabulary form: bf16 rows + partial sums (split norm), a wide projection with nothing but the f32 / argmax outputs
static bool vlt5_declin_vocab_ok(const DecLinArgs& a) {
    const int RB = (a.rows + 15) / 16;
    return DECLIN_VOCAB && a.xb && a.rs_part && !a.out_b && !a.resid && !a.relu && !a.nx_b && a.N >= 8192 && RB <= 8 && (a.K == 768 || a.K == 1024);
}
int vlt5_declin_launch(DecLinArgs a, hipStream_t st) {
    const bool af32 = a.xf != nullptr;
    if ((!a.xf && !a.xb) || !a.W || a.rows <= 0 || a.N <= 0 || (a.N & 3)) return VLT5_ERR_ARG;
    if (af32 && !a.ln_w) return VLT5_ERR_ARG;
    if (a.t_inc && a.t_ptr) return VLT5_ERR_ARG;
    if (a.rs_part && (af32 || a.rs_n <= 0 || a.rs_n > 64 || (a.rs_n & 3) || (((uintptr_t)a.rs_part) & 15))) return VLT5_ERR_ARG;
    if (a.nx_b && (!a.nx_w || !a.nx_ssq || a.relu || a.nx_parts != a.N / 16 || (a.N & 15) || (a.ld_nx & 3))) return VLT5_ERR_ARG;
    if ((a.ldx & 7) || (a.K & 63)) return VLT5_ERR_ALIGN;
    int kt, nw, nfrag;
    if (!declin_geometry(a.rows, a.N, a.K, af32, &kt, &nw, &nfrag, a.pmax ? 0 : a.force_nfrag)) return VLT5_ERR_ARG;
    if (a.force_nfrag > 0 && !a.pmax && (a.force_nfrag == 1 || a.force_nfrag == 2 || a.force_nfrag == 4)) {      // experiment switch
        int kt2, nw2;
        if (declin_split(a.K, af32, a.force_nfrag, &kt2, &nw2) && (af32 ? declin_pick<true>(kt2, a.force_nfrag) : declin_pick<false>(kt2, a.force_nfrag))) {
            nfrag = a.force_nfrag; kt = kt2; nw = nw2;
        }
    }
    a.RB = (a.rows + 15) / 16;
    a.CT = (a.N + 16 * nfrag - 1) / (16 * nfrag);
    a.ct_per_xcd = (a.CT + 7) / 8;
    declin_fn fn = af32 ? declin_pick<true>(kt, nfrag) : declin_pick<false>(kt, nfrag);
    if (!fn) return VLT5_ERR_ARG;
    // the resident form of the vocabulary projection (split norm, <= 8 row blocks): see declin_vocab_kernel
    if (vlt5_declin_vocab_ok(a)) {
        const int ktt = a.K / 64, ntl = (a.N + 31) / 32, nwv = a.RB;
        if (a.pmax && a.ptiles != ntl) return VLT5_ERR_ARG;
        declin_fn fv = ktt == 12 ? &declin_vocab_kernel<12> : &declin_vocab_kernel<16>;
        static std::atomic<unsigned long long> optin_v[2];
        const size_t lds_v = (size_t)2 * (32 * ktt / 8) * 1024;
        if (lds_v > 64 * 1024) {
            const int rc = vlt5_lds_optin((const void*)fv, 160 * 1024, optin_v[ktt == 12 ? 0 : 1]);
            if (rc) return rc;
        }
        int grid = ntl < 256 ? ntl : 256;
        hipLaunchKernelGGL(fv, dim3(grid), dim3(nwv * 64), lds_v, st, a);
        LAUNCH_CHECK();
        return VLT5_OK;
    }
    if (a.pmax && a.ptiles != a.CT) return VLT5_ERR_ARG;
    // the row-walking form for a wide norm-folded projection without a bf16 output (the vocabulary projection): the weight slice is
    // staged once per column tile
    if ((af32 || a.rs_part) && nfrag == 4 && nw <= 4 && a.RB > 1 && a.CT >= 256 && a.K == 64 * kt * nw && !a.out_b && !a.resid && !a.relu && !a.nx_b &&
        (kt == 3 || kt == 4)) {
        const size_t lds_r = (size_t)nw * kt * 4 * 16 * 128 + (size_t)nw * 256 * 4 + (size_t)5 * nw * 16 * 4;     // weight stage (the partial tiles overlay it) + norm-weight strips + sums of squares
        static std::atomic<unsigned long long> optin_r[4];
        declin_fn fr = af32 ? (kt == 3 ? &declin_rows_kernel<true, 3, 4> : &declin_rows_kernel<true, 4, 4>)
                            : (kt == 3 ? &declin_rows_kernel<false, 3, 4> : &declin_rows_kernel<false, 4, 4>);
        if (lds_r > 64 * 1024) {
            const int rc = vlt5_lds_optin((const void*)fr, 160 * 1024, optin_r[(af32 ? 0 : 2) + kt - 3]);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(fr, dim3(a.CT), dim3(nw * 64), lds_r, st, a);
        LAUNCH_CHECK();
        return VLT5_OK;
    }
    const size_t lds = declin_lds_bytes(af32, kt, nfrag, nw);
    static std::atomic<unsigned long long> optin[2][4][4];          // per instantiation: devices with the large-LDS attribute set
    if (lds > 64 * 1024) {
        const int rc = vlt5_lds_optin((const void*)fn, 160 * 1024, optin[af32 ? 1 : 0][kt - 1][nfrag - 1]);
        if (rc) return rc;
    }
    const int grid = 8 * a.ct_per_xcd * a.RB;
#ifdef DECLIN_TIMELINE
    if (g_declin_tl) { a.tl = g_declin_tl; g_declin_tl += g_declin_tl_stride; }
#endif
    hipLaunchKernelGGL(fn, dim3(grid), dim3(nw * 64), lds, st, a);
    LAUNCH_CHECK();
    return VLT5_OK;
}

// column tiles of the vocabulary projection (= argmax partials per row) for `rows` rows of a [N, K] weight
int vlt5_declin_tiles(int rows, int N, int K, int af32) {
    if (af32 == 2) {                                              // bf16 rows with the norm split off (row_ssq): the resident vocabulary form where it applies
        af32 = 0;
        DecLinArgs t;
        memset(&t, 0, sizeof t);
        t.rows = rows; t.N = N; t.K = K; t.xb = reinterpret_cast<const bf16_t*>(&t); t.rs_part = reinterpret_cast<const float*>(&t);
        if (vlt5_declin_vocab_ok(t)) return (N + 31) / 32;
    }
    int ks, nw, nfrag;
    if (!declin_geometry(rows, N, K, af32 != 0, &ks, &nw, &nfrag)) return -1;
    return (N + 16 * nfrag - 1) / (16 * nfrag);
}

int vlt5_dec_core_launch(const DecCoreArgs& a, int d_kv, hipStream_t st) {
    if (!a.q || !a.k || !a.v || !a.ctx || a.B <= 0 || a.H <= 0) return VLT5_ERR_ARG;
    if ((!a.t_ptr && (a.Tk < 1 || a.Tk > 64)) || (a.q_ld & 7) || (a.kv_sb & 7) || (a.kv_st & 7)) return VLT5_ERR_ARG;
    const int grid = (a.B * a.H + 3) / 4;
    switch (d_kv) {
        case 64: hipLaunchKernelGGL(dec_core_kernel<4>, dim3(grid), dim3(256), 0, st, a); break;
        case 32: hipLaunchKernelGGL(dec_core_kernel<2>, dim3(grid), dim3(256), 0, st, a); break;
        case 16: hipLaunchKernelGGL(dec_core_kernel<1>, dim3(grid), dim3(256), 0, st, a); break;
        default: return VLT5_ERR_ARG;
    }
    LAUNCH_CHECK();
    return VLT5_OK;
}

int vlt5_dec_io_launch(const DecIoArgs& a, int B, hipStream_t st) {
    if (B <= 0 || (!a.pmax && !a.tokens && a.emb_out)) return VLT5_ERR_ARG;
    if (a.emb_out && (a.d & 3)) return VLT5_ERR_ALIGN;
    hipLaunchKernelGGL(dec_io_kernel, dim3(B), dim3(256), 0, st, a);
    LAUNCH_CHECK();
    return VLT5_OK;
}

// ---- C ABI: the two kernels on their own (tests, other callers) ---------------------------------------------------------------
extern "C" int vlt5_decode_linear(const vlt5_decode_linear_desc* d, void* stream) {
    if (!d) return VLT5_ERR_ARG;
    DecLinArgs a;
    memset(&a, 0, sizeof a);
    a.xf = d->x_f32; a.xb = (const bf16_t*)d->x_bf16; a.ldx = d->ldx; a.ln_w = d->norm_w; a.eps = d->norm_eps;
    a.W = (const bf16_t*)d->w_bf16; a.rows = d->rows; a.N = d->N; a.K = d->K; a.alpha = d->alpha == 0.f ? 1.f : d->alpha;
    a.out_b = (bf16_t*)d->out_bf16; a.ldo = d->ld_out_bf16; a.split_col = d->out_bf16_2 ? d->split_col : 0x7fffffff;
    a.out_b2 = (bf16_t*)d->out_bf16_2; a.ldo2 = d->ld_out_bf16_2;
    a.out_f = d->out_f32; a.ldf = d->ld_out_f32; a.resid = d->resid; a.ldr = d->ld_resid; a.relu = d->relu;
    if ((a.xf && a.xb) || (!a.out_b && !a.out_f)) return VLT5_ERR_ARG;
    if (d->next_xn_bf16) {
        if (!a.out_f || !d->next_norm_w || !d->next_ssq || (a.N & 15)) return VLT5_ERR_ARG;
        if ((d->ld_next_xn & 3) || (((uintptr_t)d->next_norm_w) & 15)) return VLT5_ERR_ALIGN;
        a.nx_w = d->next_norm_w; a.nx_b = (bf16_t*)d->next_xn_bf16; a.ld_nx = d->ld_next_xn; a.nx_ssq = d->next_ssq; a.nx_parts = a.N / 16;
    }
    if (d->row_ssq) {
        if (!a.xb) return VLT5_ERR_ARG;
        a.rs_part = d->row_ssq; a.rs_n = d->n_row_ssq; a.eps = d->norm_eps;
    }
    if (a.out_b && ((a.ldo & 3) || (a.out_b2 && ((a.ldo2 & 3) || (a.split_col & 3))))) return VLT5_ERR_ALIGN;
    if ((a.out_f && (a.ldf & 3)) || (a.resid && (a.ldr & 3))) return VLT5_ERR_ALIGN;
    if (d->argmax_val || d->argmax_idx) {
        if (!d->argmax_val || !d->argmax_idx) return VLT5_ERR_ARG;
        a.pmax = d->argmax_val; a.pidx = d->argmax_idx;
        a.ptiles = vlt5_declin_tiles(a.rows, a.N, a.K, a.xf ? 1 : (d->row_ssq ? 2 : 0));
    }
    return vlt5_declin_launch(a, (hipStream_t)stream);
}
extern "C" int vlt5_decode_linear_tiles(int rows, int N, int K, int norm_folded) { return vlt5_declin_tiles(rows, N, K, norm_folded); }

extern "C" int vlt5_decode_attn(const vlt5_attn_desc* d, void* stream) {
    if (!d || d->Tq != 1 || d->causal || d->drop_p != 0.f) return VLT5_ERR_ARG;
    if (d->k_sb != d->v_sb || d->k_st != d->v_st) return VLT5_ERR_ARG;
    if (d->bias && (d->bias_q != 1 || d->bias_k < d->Tk)) return VLT5_ERR_ARG;
    DecCoreArgs a;
    memset(&a, 0, sizeof a);
    a.q = (const bf16_t*)d->q; a.q_ld = d->q_sb; a.k = (const bf16_t*)d->k; a.v = (const bf16_t*)d->v; a.kv_sb = d->k_sb; a.kv_st = d->k_st;
    a.ctx = (bf16_t*)d->ctx; a.ctx_ld = d->o_sb; a.bias = d->bias; a.bias_ld = d->bias_k; a.key_mask = d->key_mask; a.mask_ld = d->Tk;
    a.mask_value = d->mask_value; a.B = d->B; a.H = d->H; a.Tk = d->Tk;
    return vlt5_dec_core_launch(a, d->dk, (hipStream_t)stream);
}
