// EXPERIMENT (not in libvlt5_hip.so; see tools/experiments/skinny.h for the verdict).
// Row-panel ("skinny") bf16 GEMM for the few-row projections of the decoder stack (gfx950, wave64):
//
//   C[M,N] = epilogue( alpha * A[M,K] W[N,K]^T ),  A = bf16 activations, or A = bf16(RMSnorm(x) * w) computed in the prologue
//
// Reference ops being replaced: nn.Linear q/k/v/o of the decoder's HF T5Attention (self and EncDecAttention), wi / wo of
// T5DenseReluDense and the T5LayerNorm in front of them (VL-T5/src/modeling_t5_our.py:641-655 -> HF T5Block / T5LayerSelfAttention /
// T5LayerCrossAttention / T5LayerFF), at M = B*T = 400 rows.
//
// Why a second GEMM kernel.  The tiled kernel (gemm_kernel.h) stages BOTH operands through an LDS ring behind one workgroup barrier
// per 64-deep k-step; at M = 400 a launch is 12 dependent k-steps of a few tiles, i.e. a latency chain (8 us for 1.4 GFLOP).  Here a
// workgroup owns a PANEL of RM = 16 / 32 rows for its whole life: the panel (RM x K bf16, <= 96 KB) is put into LDS once, and the
// weights -- which no two waves share -- stream through WAVE-PRIVATE LDS rings filled by LDS-DMA (global_load_lds_dwordx4): every
// wave-instruction fetches 8 weight rows x 128 contiguous bytes (whole cache lines; fragment-shaped loads straight into VGPRs put
// 64 different lines into one instruction and ran at 18 B/clk/CU), lands lane-linear in the wave's ring, and is read back as MFMA
// fragments with conflict-free ds_read_b128 (XOR swizzle on the per-lane SOURCE address).  A wave only ever reads what it requested
// itself, so its own counted s_waitcnt vmcnt(n) is all the synchronisation the main loop needs: no workgroup barrier, SK_D - 1
// k-steps in flight per wave.  MFMA operands: v_mfma_f32_16x16x32_bf16 with the weight fragment as A and the activation fragment
// as B (a lane ends up with 4 consecutive n of one row m, as in gemm_kernel.h).
//
// Fused in: the T5 RMS norm of the panel (statistics in f32, one wave per 4 / 8 rows; column chunk 0 also leaves rstd and the bf16
// operand for the backward), ReLU, counter-based dropout (same element index as the tiled kernel: the backward regenerates the
// mask), f32 residual add, bf16 or f32 output.
#include "gemm_kernel.h"
#include "skinny.h"
#include <string.h>

extern vlt5gemm::TimingState vlt5_gemm_timing_state;

namespace {

unsigned long long* sk_timeline = nullptr;      // debug only (vlt5dbg_skinny_timeline), process-global

struct SkArgs {
    const bf16_t* A; int lda;
    const float* X; int ldx; const float* lnw; float eps; float* rstd_out; bf16_t* xn_out;
    const bf16_t* W; int ldw;
    void* C; int ldc; int out_f32;
    int M, N, K;
    float alpha;
    int relu; uint32_t drop_thr, drop_seed;
    const float* resid; int ldr;
    int nch, npanels;
    unsigned long long* tl;        // debug timeline (vlt5dbg_skinny_timeline): 8 x u64 per workgroup, or null
};

// 16-byte slot s of panel row r sits at r*K*2 + ((s ^ (r & mask)) << 4): the 16 lanes of a fragment read (16 rows, same slot)
// then hit 16 different slots of the 256-byte bank row
__device__ __forceinline__ int sk_mask(int K) { const int spr = K >> 3; return (spr & 15) == 0 ? 15 : ((spr & 7) == 0 ? 7 : 0); }

// SK_D = depth of a wave's weight ring in k-steps (SK_D - 1 in flight while one is consumed)
template <int FM, int FN, bool LN, int SK_D>
__global__ __launch_bounds__(256) void skinny_kernel(SkArgs p) {
    constexpr int RM = FM * 16, CN = FN * 64;
    constexpr int STEP_BYTES = FN * 2048;                    // one k-step of a wave: FN tiles of [16 weight rows][64 k]
    constexpr int NDMA = FN * 2;                             // wave-instructions per k-step (8 rows x 128 bytes each)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lg = lane >> 4;
    unsigned long long tlv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SKTL(i) do { if (p.tl && tid == 0) tlv[i] = __builtin_readcyclecounter(); } while (0)
    if (p.tl && tid == 0) tlv[7] = __builtin_amdgcn_s_memrealtime();
    SKTL(0);
    // XCD-aware order (the dispatcher places workgroup b on XCD b % 8, each XCD has a private L2): the (chunk, panel) pairs are laid
    // out chunk-major and every XCD takes one contiguous run of them, so the panels that stream the same weight rows share an L2
    // and the rows leave HBM once.  Bijective for any grid (speed only).
    int pair;
    {
        const int nt = gridDim.x, b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7, loc = b >> 3;
        pair = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int chunk = pair / p.npanels, panel = pair - chunk * p.npanels;
    const int m0 = panel * RM, nw0 = chunk * CN + wave * (FN * 16);
    const int K = p.K, nks = K >> 6, mask = sk_mask(K), rowb = K * 2;
    const int panel_bytes = (RM * rowb + 4095) & ~4095;
    char* ring = smem + panel_bytes + wave * (SK_D * STEP_BYTES);

    // ---- weight stream.  DMA instruction h of tile j: lane -> weight row 8h + (lane >> 3), physical 16-byte chunk lane & 7 of the
    // row's 128 bytes; it holds logical chunk pc ^ ((row >> 1) & 7), so that the fragment reads below (16 rows, same logical
    // chunk) spread over all 64 banks
    const bf16_t* wsrc[NDMA];
#pragma unroll
    for (int q = 0; q < NDMA; ++q) {
        const int r = (q & 1) * 8 + (lane >> 3);
        const int n = min(nw0 + (q >> 1) * 16 + r, p.N - 1);
        wsrc[q] = p.W + (size_t)n * p.ldw + (((lane & 7) ^ ((r >> 1) & 7)) << 3);
    }
    auto request = [&](int s) __attribute__((always_inline)) {          // k-step s -> ring slot s % SK_D
        char* dst = ring + (s % SK_D) * STEP_BYTES;
#pragma unroll
        for (int q = 0; q < NDMA; ++q) vlt5gemm::lds_dma16<true>(wsrc[q] + s * 64, dst + q * 1024);
    };
#pragma unroll
    for (int u = 0; u < SK_D - 1; ++u)
        if (u < nks) request(u);

    SKTL(1);
    // ---- prologue: the panel into LDS
    if constexpr (LN) {
        constexpr int RPW = RM / 4;                  // rows per wave
        constexpr int KCH = 4;                       // d_model <= 1024: this lane's columns are c = lane*4 + kk*256
        float4 wv[KCH];
#pragma unroll
        for (int kk = 0; kk < KCH; ++kk) {
            const int c = lane * 4 + kk * 256;
            wv[kk] = c < K ? *reinterpret_cast<const float4*>(p.lnw + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int rb = 0; rb < RPW; rb += 4) {
            float4 xv[4][KCH];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + wave * RPW + rb + q;
#pragma unroll
                for (int kk = 0; kk < KCH; ++kk) {
                    const int c = lane * 4 + kk * 256;
                    xv[q][kk] = (c < K && m < p.M) ? *reinterpret_cast<const float4*>(p.X + (size_t)m * p.ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = wave * RPW + rb + q, m = m0 + r;
                float ss = 0.f;
#pragma unroll
                for (int kk = 0; kk < KCH; ++kk)
                    ss += xv[q][kk].x * xv[q][kk].x + xv[q][kk].y * xv[q][kk].y + xv[q][kk].z * xv[q][kk].z + xv[q][kk].w * xv[q][kk].w;
                ss = wave_sum(ss);
                const float rs = rsqrtf(ss / (float)K + p.eps);
                if (chunk == 0 && lane == 0 && m < p.M && p.rstd_out) p.rstd_out[m] = rs;
#pragma unroll
                for (int kk = 0; kk < KCH; ++kk) {
                    const int c = lane * 4 + kk * 256;
                    if (c >= K) continue;
                    uint2 pk;
                    pk.x = pack_bf16x2(wv[kk].x * (xv[q][kk].x * rs), wv[kk].y * (xv[q][kk].y * rs));
                    pk.y = pack_bf16x2(wv[kk].z * (xv[q][kk].z * rs), wv[kk].w * (xv[q][kk].w * rs));
                    *reinterpret_cast<uint2*>(smem + r * rowb + (((c >> 3) ^ (r & mask)) << 4) + (c & 7) * 2) = pk;
                    if (chunk == 0 && p.xn_out && m < p.M) *reinterpret_cast<uint2*>(p.xn_out + (size_t)m * K + c) = pk;
                }
            }
        }
    } else {
        // LDS-DMA as well: a wave-instruction fills 64 consecutive 16-byte slots of the panel image, the XOR swizzle is applied
        // to the per-lane SOURCE address.  Rows past M repeat row M-1 (their products are never stored); the image is
        // allocated in whole 4 KB pieces, slots past the panel fetch its last slot again.
        const int spr = K >> 3, nslot = RM * spr;
        for (int base = 0; base < nslot; base += 256) {
            const int P = min(base + tid, nslot - 1);
            const int r = P / spr, ps = P - r * spr, m = min(m0 + r, p.M - 1);
            vlt5gemm::lds_dma16<true>(p.A + (size_t)m * p.lda + ((ps ^ (r & mask)) << 3), smem + (size_t)(base + (tid & ~63)) * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the panel pieces are the youngest requests: everything has landed)
    }
    SKTL(2);
    __syncthreads();
    SKTL(3);

    // ---- main loop: no workgroup barrier; k-step s + SK_D - 1 is requested into the ring slot step s - 1 has retired, then the
    // wave waits until all but the youngest SK_D - 1 k-steps have landed (memory returns in order)
    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const char* arow[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) arow[i] = smem + (i * 16 + lrow) * rowb;
    const int xr = lrow & mask;
    const int wx = (lrow >> 1) & 7;
    const int woff0 = lrow * 128 + ((lg ^ wx) << 4), woff1 = lrow * 128 + (((4 + lg) ^ wx) << 4);
    for (int s = 0; s < nks; ++s) {
        if (s + SK_D - 1 < nks) {
            asm volatile("" ::: "memory");          // (the fragment reads of step s - 1 stay above the request that overwrites their slot)
            request(s + SK_D - 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SK_D - 1) * NDMA) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const char* wt = ring + (s % SK_D) * STEP_BYTES;
        const int s0 = ((s * 8 + lg) ^ xr) << 4, s1 = ((s * 8 + 4 + lg) ^ xr) << 4;
        bf16x8_t a0[FM], a1[FM];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            a0[i] = *reinterpret_cast<const bf16x8_t*>(arow[i] + s0);
            a1[i] = *reinterpret_cast<const bf16x8_t*>(arow[i] + s1);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const bf16x8_t w0 = *reinterpret_cast<const bf16x8_t*>(wt + j * 2048 + woff0);
            const bf16x8_t w1 = *reinterpret_cast<const bf16x8_t*>(wt + j * 2048 + woff1);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, a0[i], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, a1[i], acc[i][j], 0, 0, 0);
            }
        }
    }

    SKTL(4);
    // ---- epilogue: lane holds C[m][n .. n+3], m = m0 + i*16 + (lane & 15), n = nw0 + j*16 + (lane >> 4)*4
    const float dscale = drop_scale(p.drop_thr);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + i * 16 + lrow;
        float4 rs[FN];
        if (p.resid) {
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = nw0 + j * 16 + lg * 4;
                rs[j] = (m < p.M && n < p.N) ? *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = nw0 + j * 16 + lg * 4;
            if (m >= p.M || n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * p.alpha;
            if (p.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (p.drop_thr) {
                bool kp[4];
                drop_keep4(p.drop_seed, (uint32_t)m * (uint32_t)p.N + (uint32_t)n, p.drop_thr, kp);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = kp[r] ? v[r] * dscale : 0.f;
            }
            if (p.resid) { v[0] += rs[j].x; v[1] += rs[j].y; v[2] += rs[j].z; v[3] += rs[j].w; }
            if (p.out_f32) {
                store_wt16f(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n, make_float4(v[0], v[1], v[2], v[3]));
            } else {
                uint2 pk;
                pk.x = pack_bf16x2(v[0], v[1]);
                pk.y = pack_bf16x2(v[2], v[3]);
                store_wt8(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n, pk);
            }
        }
    }
    if (p.tl && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tlv[5] = __builtin_readcyclecounter();
        tlv[6] = __builtin_amdgcn_s_memrealtime();
        for (int q = 0; q < 8; ++q) p.tl[(size_t)blockIdx.x * 8 + q] = tlv[q];
    }
#undef SKTL
}

template <int FM, int FN, bool LN, int SK_D>
int sk_launch(const SkArgs& a, int npanels, hipStream_t st) {
    // the panel (whole LDS-DMA pieces of 256 slots) + one weight ring per wave
    const size_t lds = ((size_t)FM * 16 * a.K * 2 + 4095) / 4096 * 4096 + (size_t)4 * SK_D * FN * 2048;
    if (lds > 160 * 1024) return VLT5_ERR_ARG;
    static std::atomic<unsigned long long> optin{0};
    if (int rc = vlt5_lds_optin(reinterpret_cast<const void*>(&skinny_kernel<FM, FN, LN, SK_D>), 160 * 1024, optin)) return rc;
    const_cast<SkArgs&>(a).npanels = npanels;
    const_cast<SkArgs&>(a).tl = sk_timeline;
    const dim3 grid(a.nch * npanels);
    auto& tm = vlt5_gemm_timing_state;
    if (tm.on && tm.rec.size() < tm.cap) {
        const size_t i = tm.rec.size();
        vlt5_gemm_timing_rec r;
        memset(&r, 0, sizeof r);
        r.M = a.M; r.N = a.N; r.K = a.K; r.batch = 1; r.tile_m = FM * 16; r.tile_n = FN * 64; r.b_kmajor = 0; r.a_kmajor = 0;
        r.splits = 1; r.workgroups = (int)grid.x; r.out_f32 = a.out_f32;
        tm.rec.push_back(r);
        hipExtLaunchKernelGGL((skinny_kernel<FM, FN, LN, SK_D>), grid, dim3(256), lds, st, tm.ev[2 * i], tm.ev[2 * i + 1], 0, a);
    } else {
        hipLaunchKernelGGL((skinny_kernel<FM, FN, LN, SK_D>), grid, dim3(256), lds, st, a);
    }
    LAUNCH_CHECK();
    return VLT5_OK;
}

template <int FM, bool LN>
int sk_launch_fn(const SkArgs& a, int fn, int npanels, hipStream_t st) {
    // ring depth by what the LDS holds beside the panel: the requests in flight per CU (ring bytes) over the memory latency are the
    // rate the weights arrive at -- measured with 3 k-steps in flight (48 KB per CU): 12 B/clk/CU
    const size_t room = 160 * 1024 - (((size_t)FM * 16 * a.K * 2 + 4095) / 4096 * 4096);
    if (fn == 1) return sk_launch<FM, 1, LN, 8>(a, npanels, st);                       // 64 KB of rings
    if (fn == 2) return room >= 96 * 1024 ? sk_launch<FM, 2, LN, 6>(a, npanels, st) : sk_launch<FM, 2, LN, 4>(a, npanels, st);
    return room >= 96 * 1024 ? sk_launch<FM, 3, LN, 4>(a, npanels, st) : sk_launch<FM, 3, LN, 2>(a, npanels, st);
}

}  // namespace

// debug probe (not part of the public ABI): per-workgroup phase clocks of the next launches go to buf (8 x u64 per workgroup)
extern "C" int vlt5dbg_skinny_timeline(void* buf) { sk_timeline = (unsigned long long*)buf; return VLT5_OK; }

// whether vlt5_skinny_gemm takes the shape: K a multiple of 64 with a panel of 16 rows inside the LDS, norm prologue up to d = 1024
extern "C" int vlt5_skinny_ok(int M, int N, int K, int with_norm) {
    if (M < 1 || N < 8 || K < 64 || (K & 63) || (N & 7)) return 0;
    if ((size_t)16 * K * 2 > 144 * 1024) return 0;
    if (with_norm && K > 1024) return 0;
    return 1;
}

extern "C" int vlt5_skinny_gemm(const vlt5_skinny_desc* d, void* stream) {
    if (!d || !d->W || !d->C || (!d->A && !d->ln_x)) return VLT5_ERR_ARG;
    const bool ln = d->ln_x != nullptr;
    if (ln && !d->ln_w) return VLT5_ERR_ARG;
    if (!vlt5_skinny_ok(d->M, d->N, d->K, ln)) return VLT5_ERR_ARG;
    if ((d->ldw & 7) || (!ln && (d->lda & 7)) || (ln && (d->ldx & 3)) || (d->ldc & 3) || (d->resid && (d->ldr & 3))) return VLT5_ERR_ALIGN;
    if (d->resid && !d->out_f32) return VLT5_ERR_ARG;
    SkArgs a;
    a.A = (const bf16_t*)d->A; a.lda = d->lda;
    a.X = d->ln_x; a.ldx = d->ldx; a.lnw = d->ln_w; a.eps = d->eps; a.rstd_out = d->rstd_out; a.xn_out = (bf16_t*)d->xn_out_bf16;
    a.W = (const bf16_t*)d->W; a.ldw = d->ldw;
    a.C = d->C; a.ldc = d->ldc; a.out_f32 = d->out_f32;
    a.M = d->M; a.N = d->N; a.K = d->K; a.alpha = d->alpha;
    a.relu = d->relu; a.drop_thr = d->drop_p > 0.f ? drop_thr16(d->drop_p) : 0u; a.drop_seed = d->drop_seed;
    a.resid = d->resid; a.ldr = d->ldr;
    // panel height: 32 rows while the panel fits 64 KB of LDS (two workgroups per CU), else 16
    int rm = d->panel_rows;
    if (rm != 16 && rm != 32) rm = ((size_t)32 * d->K * 2 <= 64 * 1024) ? 32 : 16;
    if ((size_t)rm * d->K * 2 > 144 * 1024) return VLT5_ERR_ARG;
    const int npanels = (d->M + rm - 1) / rm;
    // columns per workgroup (64 / 128 / 192): the fewest bytes through the busiest CU's fill path -- rounds of 256 workgroups x
    // (weight rows + panel) -- the wider chunk on a tie (fewer re-reads of the panel)
    int fn = d->chunk_cols / 64;
    if (fn < 1 || fn > 3) {
        double best = 1e30;
        for (int f = 3; f >= 1; --f) {
            const long wgs = (long)npanels * ((d->N + 64 * f - 1) / (64 * f));
            const double cost = (double)((wgs + 255) / 256) * ((double)64 * f * d->K * 2 + (double)rm * d->K * (ln ? 4 : 2));
            if (cost < best - 1e-9) { best = cost; fn = f; }
        }
    }
    a.nch = (d->N + 64 * fn - 1) / (64 * fn);
    hipStream_t st = (hipStream_t)stream;
    if (rm == 32) return ln ? sk_launch_fn<2, true>(a, fn, npanels, st) : sk_launch_fn<2, false>(a, fn, npanels, st);
    return ln ? sk_launch_fn<1, true>(a, fn, npanels, st) : sk_launch_fn<1, false>(a, fn, npanels, st);
}
