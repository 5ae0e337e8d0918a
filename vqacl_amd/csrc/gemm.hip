// bf16 MFMA GEMM for the VL-T5 projections (gfx950, wave64): argument checks, tile / split-K policy and dispatch.
// The kernel template lives in gemm_kernel.h (design notes there); its instantiations are compiled one tile shape per
// translation unit (gemm_t*.hip).
#include "gemm_kernel.h"

vlt5gemm::TimingState vlt5_gemm_timing_state;
#define g_timing vlt5_gemm_timing_state

namespace vlt5gemm {
int launch_256x256(const GemmArgs& a, int akm, int bkm, int splits, int batch, hipStream_t st);
int launch_224x256(const GemmArgs& a, int akm, int bkm, int splits, int batch, hipStream_t st);
int launch_160x256(const GemmArgs& a, int akm, int bkm, int splits, int batch, hipStream_t st);
int launch_128x128(const GemmArgs& a, int akm, int bkm, int splits, int batch, hipStream_t st);
int launch_128x64(const GemmArgs& a, int akm, int bkm, int splits, int batch, hipStream_t st);
int launch_64x128(const GemmArgs& a, int akm, int bkm, int splits, int batch, hipStream_t st);
int launch_64x64(const GemmArgs& a, int akm, int bkm, int splits, int batch, hipStream_t st);
}  // namespace vlt5gemm
using namespace vlt5gemm;

extern "C" int vlt5_gemm_bf16(vlt5_gemm_desc* d, void* stream) {
    if (!d || !d->A || !d->B || !d->C) return VLT5_ERR_ARG;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0) return VLT5_ERR_ARG;
    // contiguous dimensions are read/written as 8-element (16-byte) vectors
    int a_contig = d->a_kmajor ? d->M : d->K, b_contig = d->b_kmajor ? d->N : d->K;
    if ((a_contig & 7) || (b_contig & 7) || (d->N & 7) || (d->lda & 7) || (d->ldb & 7) || (d->ldc & 3)) return VLT5_ERR_ALIGN;
    if (d->resid && (d->ldr & 3)) return VLT5_ERR_ALIGN;
    if (d->gate && (d->ldg & 3)) return VLT5_ERR_ALIGN;
    if (!d->out_f32 && (d->ldc & 7)) return VLT5_ERR_ALIGN;           // bf16 rows are written as 16-byte vectors
    if (d->accum && !d->out_f32) return VLT5_ERR_ARG;
    if (d->gate && (d->resid || d->accum)) return VLT5_ERR_ARG;      // the epilogue holds ONE auxiliary operand per fragment
    if (d->split_k > 1 && (!d->out_f32 || !d->workspace || d->ldc != d->N || d->bias || d->relu || d->gate || d->drop_p > 0.f || d->resid))
        return VLT5_ERR_ARG;
    if (d->defer_reduce && d->accum) return VLT5_ERR_ARG;            // the consumer of the slabs adds nothing else
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GemmArgs a;
    a.A = (const bf16_t*)d->A; a.B = (const bf16_t*)d->B; a.C = d->C;
    a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
    a.alpha = d->alpha; a.bias = d->bias; a.resid = d->resid; a.ldr = d->ldr;
    a.gate = (const bf16_t*)d->gate; a.ldg = d->ldg; a.gate_scale = d->gate_scale;
    if (d->gate_bits || d->relu_bits_out) {        // ReLU sign bits: whole bytes per row, single un-split launches with the dedicated epilogues
        if ((d->N & 7) || d->ld_bits < d->N / 8 || d->batch > 1 || d->split_k > 1 || d->grouped_with) return VLT5_ERR_ARG;
        if (d->gate_bits && (d->gate || d->resid || d->accum || d->bias || d->relu || d->drop_p > 0.f || d->out_f32)) return VLT5_ERR_ARG;
        if (d->relu_bits_out && (!d->relu || d->out_f32 || d->bias || d->gate || d->gate_bits || d->resid || d->accum)) return VLT5_ERR_ARG;
    }
    a.drop_thr = d->drop_p > 0.f ? drop_thr16(d->drop_p) : 0u; a.drop_seed = d->drop_seed;
    a.relu = d->relu; a.out_f32 = d->out_f32; a.accum = d->accum;
    a.ktiles_per_split = 0; a.c_split_stride = 0;
    a.C2 = nullptr;
    if (d->c_bf16_copy) {
        // plain f32 outputs only (the weight-gradient GEMMs): not accumulated, no fused epilogue, dense rows
        if (!d->out_f32 || d->accum || d->bias || d->relu || d->gate || d->resid || d->drop_p > 0.f || d->defer_reduce) return VLT5_ERR_ARG;
        if ((((uintptr_t)d->c_bf16_copy) & 7)) return VLT5_ERR_ALIGN;
        a.C2 = (bf16_t*)d->c_bf16_copy;
    }
    a.sumsq = nullptr; a.sumsq_zstride = 0;
    if (d->sumsq) {
        if (!d->out_f32 || d->accum || d->bias || d->relu || d->gate || d->resid || d->drop_p > 0.f || d->split_k > 1) return VLT5_ERR_ARG;
        a.sumsq = d->sumsq; a.sumsq_zstride = d->sumsq_batch_stride;
    }
    a.batch_a = d->batch_stride_a; a.batch_b = d->batch_stride_b; a.batch_c = d->batch_stride_c;
    a.rs_part = nullptr; a.rs_n = 0; a.rs_inv_d = 0.f; a.rs_eps = 0.f; a.rstd_out = nullptr;
    if (d->norm_partials) {            // consumer of a folded T5 RMS norm: rows scaled by rstd from the producer's partial sums of squares
        if (d->norm_nparts < 1 || d->norm_nparts > SSQ_STRIDE || d->norm_d < 1 || d->a_kmajor || d->b_kmajor || d->split_k > 1 || d->grouped_with ||
            d->out_f32 || d->gate || d->resid || d->accum || d->bias)        // (bf16 outputs without auxiliary operand: the projections behind a norm)
            return VLT5_ERR_ARG;
        if (((uintptr_t)d->norm_partials) & 15) return VLT5_ERR_ALIGN;
        a.rs_part = d->norm_partials; a.rs_n = d->norm_nparts; a.rs_inv_d = 1.0f / (float)d->norm_d; a.rs_eps = d->norm_eps;
        a.rstd_out = d->norm_rstd_out;
    }
    a.emit_w = nullptr; a.emit_xw = nullptr; a.emit_ssq = nullptr;
    const bool emit = d->emit_xw_bf16 != nullptr;
    if (emit) {                        // producer of a folded norm: f32 residual epilogue only
        if (!d->emit_norm_w || !d->emit_partials || !d->out_f32 || !d->resid || d->accum || d->bias || d->relu || d->gate || d->split_k > 1 ||
            d->grouped_with || d->batch > 1 || d->N > 1024 || d->a_kmajor || d->b_kmajor)
            return VLT5_ERR_ARG;
        if ((((uintptr_t)d->emit_xw_bf16) & 15) || (((uintptr_t)d->emit_norm_w) & 15) || (d->ldc & 7)) return VLT5_ERR_ALIGN;
        a.emit_w = d->emit_norm_w; a.emit_xw = (bf16_t*)d->emit_xw_bf16; a.emit_ssq = d->emit_partials;
    }
    // ReLU sign bits ride in fields their epilogues do not use otherwise (GemmArgs): the gate as a bit matrix = gate pointer with a
    // NEGATIVE leading dimension; the bit matrix a ReLU bf16 epilogue writes = emit_xw with its bytes per row in ldr
    if (d->gate_bits) { a.gate = (const bf16_t*)d->gate_bits; a.ldg = -d->ld_bits; }
    if (d->relu_bits_out) { a.emit_xw = (bf16_t*)d->relu_bits_out; a.ldr = d->ld_bits; }
    a.grp_tiles = 0; a.gA = nullptr;
    const vlt5_gemm_desc* g2 = d->grouped_with;
    if (g2) {
        // the second problem of a grouped launch: same reduction, batch, operand orders and (plain f32) epilogue
        if (!g2->A || !g2->B || !g2->C || g2->K <= 0 || g2->a_kmajor != d->a_kmajor || g2->b_kmajor != d->b_kmajor ||
            g2->alpha != d->alpha || !d->out_f32 || !g2->out_f32 || d->accum || g2->accum || d->split_k > 1 || g2->split_k > 1 || d->bias ||
            g2->bias || d->relu || g2->relu || d->gate || g2->gate || d->resid || g2->resid || d->drop_p > 0.f || g2->drop_p > 0.f)
            return VLT5_ERR_ARG;
        int c2 = g2->a_kmajor ? g2->M : g2->K, b2c = g2->b_kmajor ? g2->N : g2->K;
        if ((c2 & 7) || (b2c & 7) || (g2->N & 7) || (g2->lda & 7) || (g2->ldb & 7) || (g2->ldc & 3)) return VLT5_ERR_ALIGN;
        a.gA = (const bf16_t*)g2->A; a.gB = (const bf16_t*)g2->B; a.gC = g2->C; a.gM = g2->M; a.gN = g2->N; a.gK = g2->K;
        a.grp_t2 = g2->batch > 1 ? g2->batch : 1;            // (the batch count of the second problem travels here until the tile shape is known)
        a.glda = g2->lda; a.gldb = g2->ldb; a.gldc = g2->ldc;
        a.gbatch_a = g2->batch_stride_a; a.gbatch_b = g2->batch_stride_b; a.gbatch_c = g2->batch_stride_c;
        a.gC2 = (bf16_t*)g2->c_bf16_copy; a.gsumsq = g2->sumsq; a.gsumsq_zstride = g2->sumsq_batch_stride;
    }
    const int batch = d->batch > 1 ? d->batch : 1;
    if (batch > 1 && (d->split_k > 1 || d->gate)) return VLT5_ERR_ARG;

    int bm = d->tile_m, bn = d->tile_n;
    if (bm == 0 || bn == 0) {
        // heuristic from tools/gemm_sweep.py on MI355X (profiles/): the GEMMs of this model are small -- a 4480 x 768
        // output is only 210 tiles of 128 x 128 for 256 CUs
        const int sk = (d->split_k > 1 ? d->split_k : 1) * batch;
        auto tiles = [&](int tm, int tn) {
            long t = (long)((d->M + tm - 1) / tm) * ((d->N + tn - 1) / tn);
            t *= sk;
            if (g2) t += (long)((g2->M + tm - 1) / tm) * ((g2->N + tn - 1) / tn) * (g2->batch > 1 ? g2->batch : 1);
            return t;
        };
        // 256 x 256 (8 waves, one workgroup per CU): half the LDS-fill traffic per flop of 128 x 128 -- wins once its tiles
        // fill most of the 256 CUs (wide-N forward / dgrad GEMMs with row-major A; also a small output with a very long reduction
        // cut into slices by vlt5_gemm_auto_split: the input gradients of lm_head and of the stacked cross-attention K/V projection).
        // (Round 1 found the k-major A variant slower than 128 x 128 for the weight gradients -- 455 against 385-410 us; that was the
        // compiler's wait in front of every transpose read, see below.)
        // Weight gradients (both operands k-major, reduction over the rows of the batch): since the LDS-DMA of the k-major kernels is
        // issued as assembly (gemm_kernel.h lds_dma16) their prefetch ring works, and the larger tiles win -- in situ, B = 80:
        // 768x3072x4480 x6 layers 168 -> 151 us and the stacked cross-K/V 18432x768x4640 170 -> 160 us with 256 x 256 (a long
        // reduction only: at K = 400, the decoder, its prologue / epilogue dominate); 2304x768x4480 x6 159 -> 148 us and
        // 768x768x400 x12 24 -> 18 us with 128 x 128 instead of 64 x 128.  (experiment switches: vlt5_tuning.gemm_t128_kmkm, gemm_t256_km)
        const vlt5_tuning* tn = d->tuning;
        const int t128_kmkm = tn && tn->gemm_t128_kmkm > 0 ? tn->gemm_t128_kmkm : 100;
        const int t256_km = tn && tn->gemm_t256_km > 0 ? tn->gemm_t256_km : 160;
        const int t128 = (d->a_kmajor && d->b_kmajor) ? t128_kmkm : 768;
        const int t256_min = tn && tn->gemm_t256_min > 0 ? tn->gemm_t256_min : 100;   // (160 until the t5-large shapes were measured: 1792 x 4096 x 1024 as 192 tiles of 160 x 256, +3 % on that step)
        if (d->a_kmajor && d->K >= 1024 && tiles(256, 256) >= t256_km) { bm = 256; bn = 256; }
        else if (!d->a_kmajor && (tiles(256, 256) >= t256_min || (d->K >= 8192 && d->split_k > 1 && tiles(256, 256) >= 128))) {
            // 8-wave kernel; its tile HEIGHT is chosen to fill the 256 CUs: a launch costs about (fixed part + k-steps x height/256)
            // per round of 256 workgroups, the fixed part (launch, prologue, epilogue) being worth ~9 k-steps of the full tile
            // (FFN-in forward 4480 x 3072: 216 tiles of 256 rows = 84 % of the CUs -> 240 tiles of 224 rows)
            bn = 256;
            const int nks = (d->K + BK - 1) / BK / (d->split_k > 1 ? d->split_k : 1);
            double best = 1e30;
            for (int h : {256, 224, 160}) {
                const long t = tiles(h, 256);
                const double cost = (double)((t + 255) / 256) * (9.0 + nks * (h / 256.0));
                if (cost < best - 1e-9) { best = cost; bm = h; }
            }
        }
        // (with k-major operands the 64-wide tiles run the deeper fragment pipeline, KM_STEP, and win below this threshold;
        // above it -- the layer-batched weight gradients -- 128 x 128 is still 25 % faster)
        else if (tiles(128, 128) >= t128) { bm = 128; bn = 128; }       // >= 3 workgroups per CU of the big tile
        else if (tiles(64, 128) >= 256) {                              // 3-stage ring, 2 workgroups per CU
            // input-gradient layout (row-major dY, k-major W): the tall tile reads the transpose-read operand half as often per
            // flop (sweep r01_f: 26.0 vs 27.5 us on 4480x768x2304, 171.6 vs 182.3 on 4640x768x18432)
            // (round 5, step-faithful sweep: warm, the opposite choices win by ~10 % on the 4480-row launches -- in-step A/B runs decide:
            // vlt5_tuning.gemm_rmkm_tile / gemm_rmrm_f32_tile)
            if (!d->a_kmajor && d->b_kmajor) {
                if (tn && tn->gemm_rmkm_tile == 2) { bm = 64; bn = 128; } else { bm = 128; bn = 64; }
            } else {
                bm = 64; bn = 128;
                // an f32 sublayer output: 128x64 (a producer of a folded norm's partials gets its tile from the engine, which knows
                // whether the consumer takes 24 partials per row -- two per 64-column tile -- or, the fused encoder attention kernel, 16)
                if (tn && tn->gemm_rmrm_f32_tile == 2 && !d->a_kmajor && !d->b_kmajor && d->out_f32 &&
                    !emit) { bm = 128; bn = 64; }
            }
        }
        else {
            bm = 64; bn = 64;                                          // small-M (decoder) problems: most workgroups
            // ... except a wide input gradient (decoder FFN-out dgrad 400 x 3072 x 768 with its ReLU gate): 192 tiles of 128 x 64 read
            // the transpose-read weight operand half as often (replay 8.0 against 9.5 us; vlt5_tuning.gemm_dec_tall = 1: the 64 x 64 tiles)
            const int tall = !(tn && tn->gemm_dec_tall == 1);
            if (tall && !d->a_kmajor && d->b_kmajor && d->N >= 2048 && tiles(128, 64) >= 160) { bm = 128; bn = 64; }
        }
    }
    if (emit && bn > 128) { bm = 64; bn = 128; }     // (the norm-emitting epilogue lives in the 4-wave tiles: two column slices per tile)
    if (!(((bm == 128 || bm == 64) && (bn == 128 || bn == 64)) || ((bm == 256 || bm == 224 || bm == 160) && bn == 256))) return VLT5_ERR_ARG;
    if (emit) d->emit_nparts = ((d->N + bn - 1) / bn) * 2;
    if ((bm == 224 || bm == 160) && d->a_kmajor) return VLT5_ERR_ARG;

    int splits = d->split_k > 1 ? d->split_k : 1;
    const int nk = (d->K + BK - 1) / BK;
    if (splits > nk) splits = nk;
    if (splits > 1) {
        a.ktiles_per_split = (nk + splits - 1) / splits;
        splits = (nk + a.ktiles_per_split - 1) / a.ktiles_per_split;
        a.c_split_stride = (long long)d->M * d->ldc;
        a.C = d->workspace;
        a.accum = 0;
        a.C2 = nullptr;                                    // the slab reduction writes the bf16 copy
    }
    int rc;
    if (bm == 256) rc = launch_256x256(a, d->a_kmajor, d->b_kmajor, splits, batch, st);
    else if (bm == 224) rc = launch_224x256(a, d->a_kmajor, d->b_kmajor, splits, batch, st);
    else if (bm == 160) rc = launch_160x256(a, d->a_kmajor, d->b_kmajor, splits, batch, st);
    else if (bm == 128 && bn == 128) rc = launch_128x128(a, d->a_kmajor, d->b_kmajor, splits, batch, st);
    else if (bm == 128 && bn == 64) rc = launch_128x64(a, d->a_kmajor, d->b_kmajor, splits, batch, st);
    else if (bm == 64 && bn == 128) rc = launch_64x128(a, d->a_kmajor, d->b_kmajor, splits, batch, st);
    else rc = launch_64x64(a, d->a_kmajor, d->b_kmajor, splits, batch, st);
    if (rc) return rc;
    d->split_used = splits > 1 ? splits : 1;
    if (splits > 1 && !d->defer_reduce) {
        long long n = (long long)d->M * d->ldc;
        int blocks = (int)((n / 4 + 255) / 256);
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, (const float*)d->workspace,
                           (float*)d->C, n, splits, n, d->accum, (bf16_t*)d->c_bf16_copy);
        LAUNCH_CHECK();
    }
    return VLT5_OK;
}

extern "C" int vlt5_gemm_timing_enable(int max_launches) {
    for (hipEvent_t e : g_timing.ev) (void)hipEventDestroy(e);
    g_timing.ev.clear(); g_timing.rec.clear(); g_timing.cap = 0; g_timing.on = false;
    if (max_launches <= 0) return VLT5_OK;
    g_timing.ev.resize(2 * (size_t)max_launches);
    for (auto& e : g_timing.ev) HIP_RET(hipEventCreate(&e));
    g_timing.rec.reserve(max_launches);
    g_timing.cap = (size_t)max_launches;
    g_timing.on = true;
    return VLT5_OK;
}
extern "C" int vlt5_gemm_timing_collect(vlt5_gemm_timing_rec* out, int cap) {
    const int n = (int)g_timing.rec.size();
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(g_timing.ev[2 * i + 1]) != hipSuccess) return -1;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_timing.ev[2 * i], g_timing.ev[2 * i + 1]) != hipSuccess) return -1;
        g_timing.rec[i].ms = ms;
        if (out && i < cap) out[i] = g_timing.rec[i];
    }
    g_timing.rec.clear();
    return n;
}

// split-K factor for a GEMM whose output has few tiles but a long reduction: aim at ~2 workgroups per CU (one for the
// 256 x 256 kernel), keep >= 4 (8) k-steps of 64 per slice, and stay inside the caller's slab scratch.  Only valid for
// plain f32 outputs.
extern "C" int vlt5_gemm_auto_split(int M, int N, int Kred, long long slab_bytes) {
    return vlt5_gemm_auto_split_tuned(M, N, Kred, slab_bytes, nullptr);
}
extern "C" int vlt5_gemm_auto_split_tuned(int M, int N, int Kred, long long slab_bytes, const vlt5_tuning* tuning) {
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    const long tiles256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    const int ksteps = (Kred + 63) / 64;
    int sk;
    if (Kred >= 8192 && tiles256 <= 64) {
        // very long reduction into few 256 x 256 tiles (input gradients of lm_head and of the stacked cross-attention K/V
        // projection): ~224 workgroups of the 8-wave kernel, >= 8 k-steps each
        sk = (int)((224 + tiles256 / 2) / tiles256);
        if (sk > 32) sk = 32;
        if (sk > ksteps / 8) sk = ksteps / 8;
    } else {
        // (K >= 768: the decoder's 400 x 768 x 768 projections whose consumer is a norm -- 84 tiles -> 3 x 84 workgroups of 4 k-steps:
        // 8.3 -> 4.6 us per launch against +1.8 us in the norm that sums the slabs; vlt5_tuning.gemm_split_kmin: experiment switch)
        const int kmin = tuning && tuning->gemm_split_kmin > 0 ? tuning->gemm_split_kmin : 768;
        if (Kred < kmin || tiles >= 128) return 1;
        sk = (int)((512 + tiles / 2) / tiles);
        // (cap 4 since round 5: in-step A/B 8.53 -> 8.48 ms per step against 8 -- the consumer norms sum half the slabs;
        // profiles/r05_c_ab_tile_policy.txt)
        const int cap = tuning && tuning->gemm_split_cap > 0 ? tuning->gemm_split_cap : 4;
        if (sk > cap) sk = cap;
        if (sk > ksteps / 4) sk = ksteps / 4;
    }
    while (sk > 1 && (long long)sk * M * N * 4 > slab_bytes) --sk;
    return sk < 1 ? 1 : sk;
}

extern "C" long long vlt5_gemm_workspace_bytes(int M, int ldc, int split_k) {
    return split_k > 1 ? (long long)M * ldc * 4 * split_k : 0;
}

// ---- debug probe (not part of the public ABI): raw semantics of ds_read_b64_tr_b16 -------------------------------
// Every lane supplies its own LDS element offset; the 4 returned bf16 per lane are written to out[lane*4 + j].
namespace {
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__global__ void tr_probe_kernel(const uint16_t* in, uint16_t* out, const int* addr_elems, int n_in) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < n_in && i < 8192; i += 64) lds[i] = in[i];
    __syncthreads();
    const int a = addr_elems[threadIdx.x];
    s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lds + a));
#pragma unroll
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
}  // namespace
#ifdef GEMM_TIMELINE
unsigned long long* vlt5_gemm_timeline_buf = nullptr;
extern "C" int vlt5dbg_set_timeline(void* buf) {
    vlt5_gemm_timeline_buf = (unsigned long long*)buf;
    return VLT5_OK;
}
#endif
extern "C" int vlt5dbg_tr_read(const void* in, void* out, const int* addr_elems, int n_in, void* stream) {
    hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)in, (uint16_t*)out, addr_elems, n_in);
    LAUNCH_CHECK();
    return VLT5_OK;
}
