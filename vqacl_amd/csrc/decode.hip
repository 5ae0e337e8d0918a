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

#ifndef DECLIN_PF
#define DECLIN_PF 1                // prefetch for the following launches (A/B: -DDECLIN_PF=0)
#endif
#define DECLIN_PF_W 3              // lines per lane of the next weight slice (<= 96 KB per 256-thread workgroup)
#define DECLIN_PF_KV 2             // lines per lane of the next attention core's keys | values (4 x 58 x 2 lines per workgroup)
#ifndef DECLIN_WT
#define DECLIN_WT 1                // outputs written through the L2 (A/B: -DDECLIN_WT=0 plain stores)
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
template <bool AF32, int KS, int NFRAG>
__global__ __launch_bounds__(512) void declin_kernel(const DecLinArgs a) {
    // LDS: per-wave norm-weight staging (AF32), partial tiles of every wave, per-wave partial sums of squares, argmax exchange
    __shared__ float4 red[8][NFRAG][64];
    __shared__ float wst[AF32 ? 8 : 1][AF32 ? 256 : 4];
    __shared__ float ssq_s[8][16];
    __shared__ float amax_s[NFRAG][16];
    __shared__ int aidx_s[NFRAG][16];

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
    const int kbase = w * (KS * 32) + kq * 8;
    const int n_tile = ct * (16 * NFRAG);

    // ---- every load of this wave in one burst, in the order they are needed: norm weights, activation rows, weight slice, residual ----
    // (loads return in order: the activation rows are converted while the weight slice is still on its way.  The scheduling barrier
    // keeps the compiler from sinking loads between the MFMAs to save registers -- that would turn one round trip into KS of them.)
    float4 xa[AF32 ? KS : 1][2];
    uint4 xb[AF32 ? 1 : KS];
    float4 wl = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (AF32) {
        // the wave's KS*32 norm weights: one coalesced 16-byte load per lane, handed round through the wave's LDS strip
        if (lane * 4 < KS * 32) wl = *reinterpret_cast<const float4*>(a.ln_w + w * (KS * 32) + lane * 4);
        const float* xp = a.xf + (size_t)mc * a.ldx + kbase;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            xa[ks][0] = *reinterpret_cast<const float4*>(xp + ks * 32);
            xa[ks][1] = *reinterpret_cast<const float4*>(xp + ks * 32 + 4);
        }
    } else {
        const bf16_t* xp = a.xb + (size_t)mc * a.ldx + kbase;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xb[ks] = *reinterpret_cast<const uint4*>(xp + ks * 32);
    }
    uint4 bw[KS][NFRAG];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int f = 0; f < NFRAG; ++f) {
            int n = n_tile + f * 16 + r16;
            n = n < a.N ? n : a.N - 1;
            bw[ks][f] = *reinterpret_cast<const uint4*>(a.W + (size_t)n * a.K + kbase + ks * 32);
        }
    }
    // the residual quad of the fragment this wave finishes (wave f finishes fragment f)
    float4 rpre = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const int n0 = n_tile + w * 16 + kq * 4;
        if (a.resid && w < NFRAG && m < a.rows && n0 < a.N) rpre = *reinterpret_cast<const float4*>(a.resid + (size_t)m * a.ldr + n0);
    }
    const int t_step = a.t_ptr ? *a.t_ptr : 0;
    // touch what the next kernels of the chain will read: one 4-byte load per 128-byte line, a fixed number of them per lane so that no
    // load waits for another (the values are consumed by nobody; the registers are named at the end of the kernel)
    unsigned pf[DECLIN_PF_W + DECLIN_PF_KV];
#pragma unroll
    for (int i = 0; i < DECLIN_PF_W + DECLIN_PF_KV; ++i) pf[i] = 0;
#if DECLIN_PF
    if (a.pf_w) {
        const long long nj = gridDim.x >> 3;
        const long long chunk = ((a.pf_xcd_bytes + nj - 1) / nj + 127) & ~127ll;
        const long long lo = (long long)xcd * a.pf_xcd_bytes + (long long)j * chunk;
        const long long hi_x = (long long)(xcd + 1) * a.pf_xcd_bytes;
        long long hi = lo + chunk < hi_x ? lo + chunk : hi_x;
        hi = hi < a.pf_total ? hi : a.pf_total;
#pragma unroll
        for (int i = 0; i < DECLIN_PF_W; ++i) {
            const long long o = lo + ((long long)i * blockDim.x + threadIdx.x) * 128;
            if (o < hi) pf[i] = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(a.pf_w) + o);
        }
    }
    if (a.pf_kv) {
        // dec_core workgroup x serves (sample, head) pairs 4x .. 4x+3: their Tk keys and values, one 128-byte line each (d_kv = 64)
        const int per = 2 * a.pf_Tk;
#pragma unroll
        for (int u = 0; u < DECLIN_PF_KV; ++u) {
            const int i = u * blockDim.x + threadIdx.x;
            const int gw = blockIdx.x * 4 + i / per, r = i % per;
            if (i < 4 * per && gw < a.pf_B * a.pf_H) {
                const int b = gw / a.pf_H, h = gw % a.pf_H, key = r >> 1, part = r & 1;
                pf[DECLIN_PF_W + u] = *reinterpret_cast<const unsigned*>(a.pf_kv + (size_t)b * a.pf_kv_sb + (size_t)key * a.pf_kv_st +
                                                                         part * (a.pf_H * a.pf_dk) + h * a.pf_dk);
            }
        }
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    TLS(1);

    f32x4_t acc[NFRAG];
#pragma unroll
    for (int f = 0; f < NFRAG; ++f) acc[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float ssq = 0.f;
    if constexpr (AF32) {
        if (lane * 4 < KS * 32) *reinterpret_cast<float4*>(&wst[w][lane * 4]) = wl;
        __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): the strip is this wave's own, no barrier
        __builtin_amdgcn_wave_barrier();
        TLS(2);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        bf16x8_t fx;
        if constexpr (AF32) {
            const float4 w0 = *reinterpret_cast<const float4*>(&wst[w][ks * 32 + kq * 8]);
            const float4 w1 = *reinterpret_cast<const float4*>(&wst[w][ks * 32 + kq * 8 + 4]);
            const float4 x0 = xa[ks][0], x1 = xa[ks][1];
            ssq += x0.x * x0.x + x0.y * x0.y + x0.z * x0.z + x0.w * x0.w + x1.x * x1.x + x1.y * x1.y + x1.z * x1.z + x1.w * x1.w;
            const uint4 pk = make_uint4(pack_bf16x2(x0.x * w0.x, x0.y * w0.y), pack_bf16x2(x0.z * w0.z, x0.w * w0.w),
                                        pack_bf16x2(x1.x * w1.x, x1.y * w1.y), pack_bf16x2(x1.z * w1.z, x1.w * w1.w));
            fx = __builtin_bit_cast(bf16x8_t, pk);
        } else {
            fx = __builtin_bit_cast(bf16x8_t, xb[ks]);
        }
#pragma unroll
        for (int f = 0; f < NFRAG; ++f)        // weight fragment as the A operand: lane (r16, kq) ends up with out[m = r16][n = kq*4 .. +3]
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bw[ks][f]), fx, acc[f], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    TLS(3);
    // ---- partial tiles of the NW waves meet in LDS ----------------------------------------------------------------------
#pragma unroll
    for (int f = 0; f < NFRAG; ++f) red[w][f][lane] = make_float4(acc[f][0], acc[f][1], acc[f][2], acc[f][3]);
    if constexpr (AF32) {
        ssq += __shfl_xor(ssq, 16, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (lane < 16) ssq_s[w][lane] = ssq;
    }
    __syncthreads();
    TLS(4);
    float rs = a.alpha;
    if constexpr (AF32) {
        float s = 0.f;
        for (int ww = 0; ww < NW; ++ww) s += ssq_s[ww][r16];
        rs *= rsqrtf(s / (float)a.K + a.eps);
    }
    for (int f = w; f < NFRAG; f += NW) {
        float4 v = red[0][f][lane];
        for (int ww = 1; ww < NW; ++ww) {
            const float4 o = red[ww][f][lane];
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
            if (lane < 16) { amax_s[f][lane] = best; aidx_s[f][lane] = bi; }
        }
    }
    if (a.pmax) {
        __syncthreads();
        if (threadIdx.x < 16 && rb * 16 + (int)threadIdx.x < a.rows) {
            float best = amax_s[0][threadIdx.x];
            int bi = aidx_s[0][threadIdx.x];
#pragma unroll
            for (int f = 1; f < NFRAG; ++f) {
                const float ov = amax_s[f][threadIdx.x];
                const int oi = aidx_s[f][threadIdx.x];
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            const size_t slot = (size_t)(rb * 16 + threadIdx.x) * a.CT + ct;
            a.pmax[slot] = best;
            a.pidx[slot] = bi;
        }
    }
    TLS(5);
#pragma unroll
    for (int i = 0; i < DECLIN_PF_W + DECLIN_PF_KV; ++i) asm volatile("" ::"v"(pf[i]));      // (keeps the prefetch loads; they landed long ago)
#ifdef DECLIN_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TLS(6);
#endif
}

typedef void (*declin_fn)(const DecLinArgs);
template <bool AF32, int KS>
declin_fn declin_pick_nfrag(int nfrag) {
    switch (nfrag) {
        case 1: return &declin_kernel<AF32, KS, 1>;
        case 2: return &declin_kernel<AF32, KS, 2>;
        case 3: if constexpr (KS <= 8) return &declin_kernel<AF32, KS, 3>; else return nullptr;
        case 4: if constexpr (KS <= 8) return &declin_kernel<AF32, KS, 4>; else return nullptr;
        default: return nullptr;
    }
}
template <bool AF32>
declin_fn declin_pick(int ks, int nfrag) {
    switch (ks) {
        case 1: return declin_pick_nfrag<AF32, 1>(nfrag);
        case 2: return declin_pick_nfrag<AF32, 2>(nfrag);
        case 3: return declin_pick_nfrag<AF32, 3>(nfrag);
        case 4: return declin_pick_nfrag<AF32, 4>(nfrag);
        case 6: return declin_pick_nfrag<AF32, 6>(nfrag);
        case 8: return declin_pick_nfrag<AF32, 8>(nfrag);
        case 12: if constexpr (!AF32) return declin_pick_nfrag<false, 12>(nfrag); else return nullptr;
        case 16: if constexpr (!AF32) return declin_pick_nfrag<false, 16>(nfrag); else return nullptr;
        default: return nullptr;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// attention core of one new token: one wave per (sample, head)
// ------------------------------------------------------------------------------------------------------------------------------
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
        kk[p][0] = *reinterpret_cast<const uint4*>(kb + (size_t)kc * a.kv_st);
        kk[p][1] = *reinterpret_cast<const uint4*>(kb + (size_t)kc * a.kv_st + 8);
        vv[p][0] = *reinterpret_cast<const uint4*>(vb + (size_t)kc * a.kv_st);
        vv[p][1] = *reinterpret_cast<const uint4*>(vb + (size_t)kc * a.kv_st + 8);
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
                if (a.out_tokens) a.out_tokens[(size_t)b * a.out_ld + a.out_col] = nxt;
            }
            tok_s = nxt;
        }
        __syncthreads();
        tok = tok_s;
    } else {
        tok = a.tokens[b];
    }
    if (a.emb_out) {                                               // next step's decoder input row (embedding lookup, eval: no dropout)
        long long id = tok < 0 ? 0 : (tok >= a.vocab ? a.vocab - 1 : tok);
        const float4* src = reinterpret_cast<const float4*>(a.table + (size_t)id * a.d);
        float4* dst = reinterpret_cast<float4*>(a.emb_out + (size_t)b * a.d);
        for (int i = tid; i < a.d / 4; i += 256) dst[i] = src[i];
    }
    if (a.bias_out && b == 0) {                                    // bias row of query position tq against keys 0..tq: [H][bias_ld]
        const int n = a.H * (a.tq + 1);
        for (int i = tid; i < n; i += 256) {
            const int h = i / (a.tq + 1), jk = i % (a.tq + 1);
            a.bias_out[(size_t)h * a.bias_ld + jk] = a.rel_table[(size_t)a.lut[(size_t)a.tq * a.lut_ld + jk] * a.H + h];
        }
    }
}

}  // namespace

// K = 32 * KS * NW with KS out of the instantiated set and NW <= 8 waves; prefers four waves
static bool declin_split(int K, bool af32, int* ks, int* nw) {
    if (K <= 0 || (K & 31)) return false;
    const int steps = K / 32;
#ifndef DECLIN_WAVES
#define DECLIN_WAVES 4             // preferred number of waves per workgroup (A/B: -DDECLIN_WAVES=8)
#endif
    const int tries[4] = {DECLIN_WAVES, DECLIN_WAVES == 4 ? 8 : 4, 2, 1};
    for (int t = 0; t < 4; ++t) {
        const int w = tries[t];
        if (steps % w) continue;
        const int s = steps / w;
        const bool ok = s == 1 || s == 2 || s == 3 || s == 4 || s == 6 || s == 8 || (!af32 && (s == 12 || s == 16));
        if (ok) { *ks = s; *nw = w; return true; }
    }
    return false;
}

// the launch geometry of (rows, N, K): k-steps per wave, waves, fragments per column tile
static bool declin_geometry(int rows, int N, int K, bool af32, int* ks, int* nw, int* nfrag) {
    if (!declin_split(K, af32, ks, nw)) return false;
    const int RB = (rows + 15) / 16;
    const int fmax = *ks >= 12 ? 2 : 4;
    for (int f = fmax; f >= 1; --f) {
        const int ctf = (N + 16 * f - 1) / (16 * f);
        if (ctf * RB >= 200 || f == 1) { *nfrag = f; return true; }
    }
    return false;
}
long long vlt5_declin_xcd_bytes(int rows, int N, int K, int af32) {
    int ks, nw, nfrag;
    if (!declin_geometry(rows, N, K, af32 != 0, &ks, &nw, &nfrag)) return 0;
    const int CT = (N + 16 * nfrag - 1) / (16 * nfrag);
    return (long long)((CT + 7) / 8) * 16 * nfrag * K * 2;
}

extern "C" int vlt5_decode_linear_supported(int K, int af32) {
    int ks, nw;
    return declin_split(K, af32 != 0, &ks, &nw) ? 1 : 0;
}

#ifdef DECLIN_TIMELINE
static long long* g_declin_tl = nullptr;       // instrumented builds only (never the product library): next stamp block
static long long g_declin_tl_stride = 0;
extern "C" void vlt5_declin_timeline(long long* buf, long long stride_per_launch) { g_declin_tl = buf; g_declin_tl_stride = stride_per_launch; }
#endif
int vlt5_declin_launch(DecLinArgs a, hipStream_t st) {
    const bool af32 = a.xf != nullptr;
    if ((!a.xf && !a.xb) || !a.W || a.rows <= 0 || a.N <= 0 || (a.N & 3)) return VLT5_ERR_ARG;
    if (af32 && !a.ln_w) return VLT5_ERR_ARG;
    if ((a.ldx & 7) || (a.K & 31)) return VLT5_ERR_ALIGN;
    int ks, nw, nfrag;
    // widest column tile that still fills the chip (>= ~200 workgroups); the wide reductions (KS >= 12) only come one or two fragments wide
    if (!declin_geometry(a.rows, a.N, a.K, af32, &ks, &nw, &nfrag)) return VLT5_ERR_ARG;
    a.RB = (a.rows + 15) / 16;
    a.CT = (a.N + 16 * nfrag - 1) / (16 * nfrag);
    a.ct_per_xcd = (a.CT + 7) / 8;
    declin_fn fn = af32 ? declin_pick<true>(ks, nfrag) : declin_pick<false>(ks, nfrag);
    if (!fn) return VLT5_ERR_ARG;
    if (a.pmax && a.ptiles != a.CT) return VLT5_ERR_ARG;
    const int grid = 8 * a.ct_per_xcd * a.RB;
#ifdef DECLIN_TIMELINE
    if (g_declin_tl) { a.tl = g_declin_tl; g_declin_tl += g_declin_tl_stride; }
#endif
    hipLaunchKernelGGL(fn, dim3(grid), dim3(nw * 64), 0, st, a);
    LAUNCH_CHECK();
    return VLT5_OK;
}

// column tiles of the vocabulary projection (= argmax partials per row) for `rows` rows of a [N, K] weight
int vlt5_declin_tiles(int rows, int N, int K, int af32) {
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
    if (a.out_b && ((a.ldo & 3) || (a.out_b2 && ((a.ldo2 & 3) || (a.split_col & 3))))) return VLT5_ERR_ALIGN;
    if ((a.out_f && (a.ldf & 3)) || (a.resid && (a.ldr & 3))) return VLT5_ERR_ALIGN;
    if (d->argmax_val || d->argmax_idx) {
        if (!d->argmax_val || !d->argmax_idx) return VLT5_ERR_ARG;
        a.pmax = d->argmax_val; a.pidx = d->argmax_idx;
        a.ptiles = vlt5_declin_tiles(a.rows, a.N, a.K, a.xf != nullptr);
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
