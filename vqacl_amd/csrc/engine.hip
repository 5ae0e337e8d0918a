// Whole-path engine: composes the kernels into VLT5 encoder/decoder forward and backward.
// Pure host code (no kernels here): each entry point only enqueues work on the caller's stream, never allocates
// and never synchronises, so one C call per phase keeps the host off the critical path.
//
// Reference being replaced: JointEncoder.forward (VL-T5/src/modeling_t5_our.py:175-339), VLT5.forward (:514-713),
// the HF T5Stack/T5Block they call, and autograd's backward of both (src/vqacl.py:461).
#include <cstdlib>
#include "common.h"
#include "vlt5_hip.h"
#include "decode.h"
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

namespace {

constexpr int MAXL = 48;

// dropout sites (each gets its own seed so masks are independent)
enum { SITE_ENC_EMBED = 1, SITE_DEC_EMBED = 2, SITE_ENC_FINAL = 5, SITE_DEC_FINAL = 6, SITE_ENC_BASE = 16, SITE_DEC_BASE = 1024 };
enum { E_PROBS = 0, E_ATTN_OUT = 1, E_FFN_H = 2, E_FFN_OUT = 3 };
enum { D_SPROBS = 0, D_SOUT = 1, D_CPROBS = 2, D_COUT = 3, D_FFN_H = 4, D_FFN_OUT = 5 };

struct Entry { std::string name; long long off; int rows, cols, bucket, decay, used; };

struct Layout {
    std::vector<Entry> e;
    long long total = 0;
    int nbuckets = 0;
    long long dec_final_ln = 0, dec_rel = 0, cross_kv = 0, enc_final_ln = 0, enc_rel = 0;
    struct DecL { long long wo, wi, ln_f, co, cq, ln_c, so, sqkv, ln_s; } dec[MAXL];
    struct EncL { long long wo, wi, ln_f, so, sqkv, ln_s; } enc[MAXL];
    long long vis_wf = 0, vis_bf = 0, vis_lnf = 0, vis_wp = 0, vis_bp = 0, vis_lnp = 0, vis_img = 0, shared = 0;
};

// Flat parameter order = order in which gradients complete during backward (decoder top -> encoder bottom -> embeddings),
// every tensor aligned to 64 elements.  q,k,v of a self-attention are adjacent (one fused [3*inner, d] GEMM operand);
// the cross-attention k,v of ALL decoder layers are adjacent (one [Ld*2*inner, d] operand over the encoder output).
void build_layout(const vlt5_config& c, Layout& L, bool names) {
    const int d = c.d_model, inner = c.num_heads * c.d_kv, ff = c.d_ff;
    long long off = 0;
    auto add = [&](const std::string& name, int rows, int cols, int bucket, int used = 1) -> long long {
        long long o = off;
        if (names) {
            int decay = (name.find("bias") != std::string::npos || name.find("LayerNorm.weight") != std::string::npos) ? 0 : 1;
            L.e.push_back({name, o, rows, cols, bucket, decay, used});
        }
        long long n = (long long)rows * (cols > 0 ? cols : 1);
        off += (n + 63) / 64 * 64;
        return o;
    };
    auto nm = [&](const char* fmt, int i) { char b[160]; snprintf(b, sizeof b, fmt, i); return std::string(b); };
    // The (tiny) norm weights are placed together in the LAST bucket: their gradients are reduced once per backward phase
    // by a single multi-job launch, i.e. later than the matrices of their layer.
    std::vector<std::pair<long long*, std::string>> norms;
    auto norm = [&](long long* slot, const std::string& name) { norms.push_back({slot, name}); };
    int bucket = 0;
    const int Ld = c.num_decoder_layers, Le = c.num_layers;
    norm(&L.dec_final_ln, "decoder.final_layer_norm.weight");
    for (int i = Ld - 1; i >= 0; --i) {
        auto& D = L.dec[i];
        D.wo = add(nm("decoder.block.%d.layer.2.DenseReluDense.wo.weight", i), d, ff, bucket);
        if (c.gated_act) {        // t5-v1.1: wi_0 | wi_1 adjacent = one [2*ff, d] GEMM operand
            D.wi = add(nm("decoder.block.%d.layer.2.DenseReluDense.wi_0.weight", i), ff, d, bucket);
            add(nm("decoder.block.%d.layer.2.DenseReluDense.wi_1.weight", i), ff, d, bucket);
        } else {
            D.wi = add(nm("decoder.block.%d.layer.2.DenseReluDense.wi.weight", i), ff, d, bucket);
        }
        norm(&D.ln_f, nm("decoder.block.%d.layer.2.layer_norm.weight", i));
        D.co = add(nm("decoder.block.%d.layer.1.EncDecAttention.o.weight", i), d, inner, bucket);
        D.cq = add(nm("decoder.block.%d.layer.1.EncDecAttention.q.weight", i), inner, d, bucket);
        norm(&D.ln_c, nm("decoder.block.%d.layer.1.layer_norm.weight", i));
        D.so = add(nm("decoder.block.%d.layer.0.SelfAttention.o.weight", i), d, inner, bucket);
        D.sqkv = add(nm("decoder.block.%d.layer.0.SelfAttention.q.weight", i), inner, d, bucket);
        add(nm("decoder.block.%d.layer.0.SelfAttention.k.weight", i), inner, d, bucket);
        add(nm("decoder.block.%d.layer.0.SelfAttention.v.weight", i), inner, d, bucket);
        norm(&D.ln_s, nm("decoder.block.%d.layer.0.layer_norm.weight", i));
        if (i == 0) L.dec_rel = add("decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", c.rel_buckets, c.num_heads, bucket);
        ++bucket;
    }
    for (int i = 0; i < Ld; ++i) {
        long long o = add(nm("decoder.block.%d.layer.1.EncDecAttention.k.weight", i), inner, d, bucket);
        if (i == 0) L.cross_kv = o;
        add(nm("decoder.block.%d.layer.1.EncDecAttention.v.weight", i), inner, d, bucket);
    }
    ++bucket;
    norm(&L.enc_final_ln, "encoder.final_layer_norm.weight");
    for (int i = Le - 1; i >= 0; --i) {
        auto& E = L.enc[i];
        E.wo = add(nm("encoder.block.%d.layer.1.DenseReluDense.wo.weight", i), d, ff, bucket);
        if (c.gated_act) {
            E.wi = add(nm("encoder.block.%d.layer.1.DenseReluDense.wi_0.weight", i), ff, d, bucket);
            add(nm("encoder.block.%d.layer.1.DenseReluDense.wi_1.weight", i), ff, d, bucket);
        } else {
            E.wi = add(nm("encoder.block.%d.layer.1.DenseReluDense.wi.weight", i), ff, d, bucket);
        }
        norm(&E.ln_f, nm("encoder.block.%d.layer.1.layer_norm.weight", i));
        E.so = add(nm("encoder.block.%d.layer.0.SelfAttention.o.weight", i), d, inner, bucket);
        E.sqkv = add(nm("encoder.block.%d.layer.0.SelfAttention.q.weight", i), inner, d, bucket);
        add(nm("encoder.block.%d.layer.0.SelfAttention.k.weight", i), inner, d, bucket);
        add(nm("encoder.block.%d.layer.0.SelfAttention.v.weight", i), inner, d, bucket);
        norm(&E.ln_s, nm("encoder.block.%d.layer.0.layer_norm.weight", i));
        if (i == 0) L.enc_rel = add("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", c.rel_buckets, c.num_heads, bucket);
        ++bucket;
    }
    for (auto& pr : norms) *pr.first = add(pr.second, d, 0, bucket);
    const char* ve = "encoder.visual_embedding.";
    L.vis_wf = add(std::string(ve) + "feat_embedding.0.weight", d, c.feat_dim, bucket);
    L.vis_bf = add(std::string(ve) + "feat_embedding.0.bias", d, 0, bucket);
    L.vis_lnf = add(std::string(ve) + "feat_embedding.1.weight", d, 0, bucket);
    L.vis_wp = add(std::string(ve) + "absolute_vis_pos_embedding.0.weight", d, 5, bucket);
    L.vis_bp = add(std::string(ve) + "absolute_vis_pos_embedding.0.bias", d, 0, bucket);
    L.vis_lnp = add(std::string(ve) + "absolute_vis_pos_embedding.1.weight", d, 0, bucket);
    L.vis_img = add(std::string(ve) + "img_order_embedding.weight", c.n_images, d, bucket);
    L.shared = add("shared.weight", c.vocab, d, bucket);
    ++bucket;
    L.nbuckets = bucket;
    // never used in forward (src/modeling_t5_our.py:379-380): kept for state_dict compatibility, excluded from buckets
    add("prototype_fc1.weight", d, d, -1, 0);
    add("prototype_fc1.bias", d, 0, -1, 0);
    add("prototype_fc2.weight", d, d, -1, 0);
    add("prototype_fc2.bias", d, 0, -1, 0);
    L.total = off;
}

struct Plan {
    int B, L, V, T, S, Sx, M, Mx, Md;
    size_t total;
    size_t feats_bf16, boxes_g, visG, vis_rf, vis_rp, mask, enc_bias;
    size_t x[2 * MAXL + 1], xr[2 * MAXL + 1];
    size_t xn_a[MAXL], qkv[MAXL], lse[MAXL], ctx[MAXL], xn_f[MAXL], h[MAXL], u[MAXL];     // u: gated FFN pre-activations [M, 2*ff]
    size_t enc_out, enc_ext, mask_ext;
    size_t dec_ids, dec_bias, kv_all;
    size_t y[3 * MAXL + 1], yr[3 * MAXL + 1];
    size_t hbits[MAXL];                    // ReLU sign bits of the encoder's FFN activations ([M][ff/8] bytes; vlt5_gemm_desc.relu_bits_out)
    size_t yn_a[MAXL], dqkv_s[MAXL], lse_s[MAXL], ctx_s[MAXL], yn_c[MAXL], qc[MAXL], lse_c[MAXL], ctx_c[MAXL], yn_f[MAXL], hd[MAXL], ud[MAXL];
    size_t dec_out, logits, lse_ce, loss_tok, row_w, loss;
    size_t ssq_e[2 * MAXL + 1], ssq_d[3 * MAXL + 1];      // per-norm partial sums of squares of the residual stream (norm folded around its GEMMs)
    // backward scratch
    size_t dx, tmp, embed_scratch, dctx, dkv_all, d_enc_ext, dS_enc, dS_dec, dlogits, slab, ln_partial, vis_partial, rel_scratch, vis_dG, small, slab2;
    // per-layer gradient operands kept until the end of the phase: the weight-gradient GEMMs of all layers run as ONE
    // batched launch per weight kind (grid.z = layer)
    size_t e_dyd_f[MAXL], e_dh[MAXL], e_dyd_a[MAXL], e_dqkv[MAXL];
    size_t d_dyd_f[MAXL], d_dh[MAXL], d_dyd_c[MAXL], d_dq_c[MAXL], d_dyd_s[MAXL], d_dqkv[MAXL];
    size_t slab_bytes;
};

void make_plan(const vlt5_config& c, int B, int L, int V, int T, Plan& p) {
    const size_t d = c.d_model, inner = (size_t)c.num_heads * c.d_kv, ff = c.d_ff, H = c.num_heads;
    const size_t ffw = c.gated_act ? 2 * ff : ff;          // width of the FFN input projection (gated: wi_0 | wi_1)
    const int Le = c.num_layers, Ld = c.num_decoder_layers;
    p.B = B; p.L = L; p.V = V; p.T = T; p.S = L + V; p.Sx = p.S + 2;
    p.M = B * p.S; p.Mx = B * p.Sx; p.Md = B * T;
    const size_t M = p.M, Mx = p.Mx, Md = p.Md;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    p.feats_bf16 = take((size_t)B * V * c.feat_dim * 2);
    p.boxes_g = take((size_t)B * V * 4 * 4);
    p.visG = take((size_t)B * V * d * 4);
    p.vis_rf = take((size_t)B * V * 4);
    p.vis_rp = take((size_t)B * V * 4);
    p.mask = take((size_t)B * p.S * 4);
    p.enc_bias = take(H * L * L * 4);
    for (int i = 0; i <= 2 * Le; ++i) { p.x[i] = take(M * d * 4); p.xr[i] = take(M * 4); }
    for (int l = 0; l < Le; ++l) {
        p.xn_a[l] = take(M * d * 2); p.qkv[l] = take(M * 3 * inner * 2); p.lse[l] = take((size_t)B * H * p.S * 4);
        p.ctx[l] = take(M * inner * 2); p.xn_f[l] = take(M * d * 2); p.h[l] = take(M * ff * 2);
        p.hbits[l] = (!c.gated_act && (ff & 7) == 0) ? take(M * (ff / 8)) : 0;
        p.u[l] = c.gated_act ? take(M * ffw * 2) : 0;
    }
    for (int i = 0; i <= 2 * Le; ++i) p.ssq_e[i] = take(M * 32 * 4);
    p.enc_out = take(Mx * d * 4);
    p.enc_ext = take(Mx * d * 2);
    p.mask_ext = take((size_t)B * p.Sx * 4);
    p.dec_ids = take(Md * 8);
    p.dec_bias = take(H * T * T * 4);
    p.kv_all = take(Mx * (size_t)Ld * 2 * inner * 2);
    for (int i = 0; i <= 3 * Ld; ++i) { p.y[i] = take(Md * d * 4); p.yr[i] = take(Md * 4); }
    for (int l = 0; l < Ld; ++l) {
        p.yn_a[l] = take(Md * d * 2); p.dqkv_s[l] = take(Md * 3 * inner * 2); p.lse_s[l] = take((size_t)B * H * T * 4);
        p.ctx_s[l] = take(Md * inner * 2); p.yn_c[l] = take(Md * d * 2); p.qc[l] = take(Md * inner * 2);
        p.lse_c[l] = take((size_t)B * H * T * 4); p.ctx_c[l] = take(Md * inner * 2); p.yn_f[l] = take(Md * d * 2);
        p.hd[l] = take(Md * ff * 2);
        p.ud[l] = c.gated_act ? take(Md * ffw * 2) : 0;
    }
    for (int i = 0; i <= 3 * Ld; ++i) p.ssq_d[i] = take(Md * 32 * 4);       // (everything whose size depends on T lies behind the encoder's part)
    p.dec_out = take(Md * d * 2);
    p.logits = take(Md * (size_t)c.vocab * 4);
    p.lse_ce = take(Md * 4); p.loss_tok = take(Md * 4); p.row_w = take(Md * 4); p.loss = take(256);
    const size_t Mmax = Mx > Md ? Mx : Md;
    p.dx = take(Mmax * d * 4);
    p.tmp = take(Mmax * d * 4);
    p.embed_scratch = take((size_t)vlt5_embed_bwd_scratch_bytes(B, L > T ? L : T, (int)d));   // (its own range: larger than tmp when V is tiny)
    p.dctx = take(Mmax * (inner > (c.gated_act ? ff : 0) ? inner : ff) * 2);     // (gated FFN: also the hidden-gradient scratch [M, ff])
    for (int l = 0; l < Le; ++l) {
        p.e_dyd_f[l] = take(M * d * 2); p.e_dh[l] = take(M * ffw * 2); p.e_dyd_a[l] = take(M * d * 2);
        p.e_dqkv[l] = take(M * 3 * inner * 2);
    }
    for (int l = 0; l < Ld; ++l) {
        p.d_dyd_f[l] = take(Md * d * 2); p.d_dh[l] = take(Md * ffw * 2); p.d_dyd_c[l] = take(Md * d * 2);
        p.d_dq_c[l] = take(Md * inner * 2); p.d_dyd_s[l] = take(Md * d * 2); p.d_dqkv[l] = take(Md * 3 * inner * 2);
    }
    p.dkv_all = take(Mx * (size_t)Ld * 2 * inner * 2);
    p.d_enc_ext = take(Mx * d * 4);
    p.dS_enc = take((size_t)Le * B * H * L * L * 4);
    p.dS_dec = take((size_t)Ld * B * H * T * T * 4);
    p.dlogits = take(Md * (size_t)c.vocab * 2);
    size_t wmax = ffw * d;
    if (3 * inner * d > wmax) wmax = 3 * inner * d;
    p.slab_bytes = 8 * wmax * 4;
    p.slab = take(p.slab_bytes);
    p.slab2 = take(p.slab_bytes);                            // split-K scratch of the weight-gradient side stream
    p.ln_partial = take((size_t)64 * LNB_MAXBLK * d * 4);          // 64 norm slots x <= 320 workgroup partials
    p.vis_partial = take(((size_t)256 * 10 * d + 2 * (size_t)B * V) * 4);   // <= 256 row splits (vlt5_vis_embed_bwd_blocks)
    size_t rs = 64 * H * (size_t)(L > T ? L : T) * (L > T ? L : T) * 4;
    p.rel_scratch = take(rs);
    p.vis_dG = take((size_t)B * V * d * 2);
    p.small = take(10 * d * 4);
    p.total = off;
}

// A T5 RMS norm folded around its GEMMs: what the consumer needs (partials per row of the producer's epilogue, where rstd goes)
struct NormIn { const float* part = nullptr; int n = 0; float* rstd_out = nullptr; };

struct Ctx {
    const vlt5_config& c;
    const vlt5_step& s;
    hipStream_t st;
    Layout lay;
    Plan p;
    char* ws;
    const float* P;
    const bf16_t* Pb;
    float* Gr;
    float pdrop;
    int d, inner, ff, H;
    mutable int ln_jobs = 0;
    mutable void* last_wait = nullptr;
    mutable long long ln_out[64];
    mutable int ln_nblk[64];
    hipStream_t side = nullptr;        // optional second stream of the backward phases (vlt5_step.side_stream)
    vlt5_tuning tun;                   // experiment switches of this call (vlt5_step.tuning, all zero = defaults)
    Ctx(const vlt5_config& c_, const vlt5_step& s_, void* stream) : c(c_), s(s_), st((hipStream_t)stream) {
        if (s.side_stream && s.side_events && s.n_side_events >= 4) side = (hipStream_t)s.side_stream;
        if (s.tuning) tun = *s.tuning; else memset(&tun, 0, sizeof tun);
        build_layout(c, lay, false);
        make_plan(c, s.B, s.L, s.V, s.T, p);
        ws = (char*)s.workspace;
        P = s.params; Pb = (const bf16_t*)s.params_bf16; Gr = s.grads;
        pdrop = s.training ? c.dropout : 0.f;
        d = c.d_model; inner = c.num_heads * c.d_kv; ff = c.d_ff; H = c.num_heads;
    }
    template <class T> T* w(size_t off) const { return reinterpret_cast<T*>(ws + off); }
    // The weight-gradient GEMMs depend on the input-gradient chain but nothing in the chain depends on them, and the chain is full
    // of latency-/HBM-bound kernels (norms, attention, few-tile GEMMs) that leave the matrix cores idle: with a side stream the
    // batched weight gradients run beside the rest of the chain.  fork(e): the side stream continues from this point of the main
    // stream; join(e): the main stream waits for everything enqueued on the side stream; on_side(): the same context, launching
    // on the side stream with its own split-K scratch.  Without a side stream all three are no-ops / the identity.
    int fork(int e) const {
        if (!side) return VLT5_OK;
        HIP_RET(hipEventRecord((hipEvent_t)s.side_events[e], st));
        HIP_RET(hipStreamWaitEvent(side, (hipEvent_t)s.side_events[e], 0));
        return VLT5_OK;
    }
    int join(int e) const {
        if (!side) return VLT5_OK;
        HIP_RET(hipEventRecord((hipEvent_t)s.side_events[e], side));
        HIP_RET(hipStreamWaitEvent(st, (hipEvent_t)s.side_events[e], 0));
        return VLT5_OK;
    }
    Ctx on_side() const {
        Ctx k2 = *this;
        if (side) { k2.st = side; k2.p.slab = p.slab2; }
        return k2;
    }
    const float* boxes() const { return s.feat_store ? w<float>(p.boxes_g) : s.boxes; }    // gathered from the store, or the caller's
    uint32_t seed(uint32_t site) const { return site_seed(s.seed, site); }
    int check(bool bwd) const {
        if (!s.params || !s.params_bf16 || !s.workspace) return VLT5_ERR_ARG;
        if (bwd && !s.grads) return VLT5_ERR_ARG;
        if (s.workspace_bytes < (long long)p.total) return VLT5_ERR_ARG;
        if (c.num_layers > MAXL || c.num_decoder_layers > MAXL) return VLT5_ERR_ARG;
        if (s.feat_store && (!s.box_store || !s.feat_slots || s.n_slots <= 0)) return VLT5_ERR_ARG;
        if (p.Sx > 64 || s.T > 64 || c.d_kv > 64 || s.B < 1 || s.L < 1 || s.V < 1 || s.T < 1) return VLT5_ERR_ARG;
        if ((d & 7) || (inner & 7) || (ff & 7) || (c.vocab & 7) || (c.feat_dim & 7) || (c.d_kv & 7)) return VLT5_ERR_ALIGN;
        return VLT5_OK;
    }

    // A T5 RMS norm folded around its GEMMs (vlt5_gemm_desc.emit_* / norm_*): the producer of the residual stream leaves
    // bf16(x * w_norm) and per-row partial sums of squares, the consumer scales its output rows by rstd and stores rstd
    // y[M,N] = epi(alpha * x[M,K] W[N,K]^T)
    int lin_fwd(const bf16_t* X, const bf16_t* W, void* C, int M, int N, int K, int out_f32, float alpha = 1.f,
                const float* bias = nullptr, int relu = 0, float dp = 0.f, uint32_t dseed = 0, const float* resid = nullptr,
                NormIn nin = NormIn(), void* bits_out = nullptr) const {
        vlt5_gemm_desc g;
        memset(&g, 0, sizeof g);
        g.tuning = &tun;
        g.A = X; g.B = W; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N;
        g.alpha = alpha; g.bias = bias; g.relu = relu; g.drop_p = dp; g.drop_seed = dseed; g.resid = resid; g.ldr = N;
        g.out_f32 = out_f32;
        if (nin.part) { g.norm_partials = nin.part; g.norm_nparts = nin.n; g.norm_d = K; g.norm_eps = c.eps; g.norm_rstd_out = nin.rstd_out; }
        if (bits_out) { g.relu_bits_out = bits_out; g.ld_bits = N / 8; }
        return vlt5_gemm_bf16(&g, st);
    }
    // y = resid + dropout(x W^T), and for the norm that consumes y next: bf16(y * w_norm) -> xw, partial sums of squares -> ssq;
    // *nparts = partials per row
    int lin_fwd_emit(const bf16_t* X, const bf16_t* W, float* Y, int M, int N, int K, float dp, uint32_t dseed, const float* resid,
                     long long w_norm, void* xw, float* ssq, int* nparts, bool wide_consumer = false) const {
        vlt5_gemm_desc g;
        memset(&g, 0, sizeof g);
        g.tuning = &tun;
        // (vlt5_tuning.gemm_rmrm_f32_tile = 2: 128 x 64 tiles where the consumer takes the 24 partials per row they leave)
        if (tun.gemm_rmrm_f32_tile == 2 && wide_consumer && (long)((M + 63) / 64) * ((N + 127) / 128) >= 256 && 2 * ((N + 63) / 64) <= 32) {
            g.tile_m = 128; g.tile_n = 64;
        }
        g.A = X; g.B = W; g.C = Y; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N;
        g.alpha = 1.f; g.drop_p = dp; g.drop_seed = dseed; g.resid = resid; g.ldr = N; g.out_f32 = 1;
        g.emit_norm_w = P + w_norm; g.emit_xw_bf16 = xw; g.emit_partials = ssq;
        int rc = vlt5_gemm_bf16(&g, st);
        *nparts = g.emit_nparts;
        return rc;
    }
    // which norms are folded (same answer in forward and backward: a function of the configuration, the shapes and the tuning
    // record only).  vlt5_tuning.fold_norm = 1 / fold_norm_dec = 2: experiment switches (A/B runs, tests of the other path)
    bool fold_on() const { return tun.fold_norm != 1 && d <= 1024 && (d & 31) == 0; }
    // (decoder: the norms in front of the cross-attention query and the FFN could lose their launches too, but their producers --
    // the 400-row attention output projections -- then run un-split over K and the 64 x 64 consumers pay the row scale: measured
    // +0.13 ms per step against -0.17 ms of norm launches saved ... and +0.13 ms of slower 84-tile GEMMs: a net loss, off by default)
    bool fold_dec() const { return tun.fold_norm_dec == 2 && fold_on(); }
    bool fold_enc_first(int l) const { return fold_on() && l > 0 && pick_split(p.M, d, ff) <= 1; }     // the norm in front of layer l's attention
    // the encoder's FFN hidden gradient gates by ReLU sign bits (1.7 MB per layer at B = 80) instead of the saved bf16 activation (27.5 MB,
    // read cold): same predicate, bit-identical gradients; vlt5_tuning.ffn_gate_bits = 1 keeps the activation as the gate
    bool bits_on() const { return tun.ffn_gate_bits != 1 && !c.gated_act && (ff & 7) == 0; }
    int pick_split(int M, int N, int Kred) const { return vlt5_gemm_auto_split_tuned(M, N, Kred, (long long)p.slab_bytes, &tun); }
    int ffw() const { return c.gated_act ? 2 * ff : ff; }
    // hidden activation of an FFN: h = dropout(act(xn Wi^T)).  ReLU: one GEMM with the activation in its epilogue.  Gated GELU (HF
    // T5DenseGatedActDense): u = xn [wi_0; wi_1]^T kept for the backward, then h = dropout(gelu_new(u0) * u1)
    // bits (optional): the ReLU sign bits of h for the backward's gate (see bits_on())
    int ffn_hidden(const bf16_t* xn, long long wi, bf16_t* u, bf16_t* h, int M, float dp, uint32_t dseed, NormIn nin = NormIn(),
                   void* bits = nullptr) const {
        if (!c.gated_act) return lin_fwd(xn, Pb + wi, h, M, ff, d, 0, 1.f, nullptr, 1, dp, dseed, nullptr, nin, bits);
        int rc = lin_fwd(xn, Pb + wi, u, M, 2 * ff, d, 0, 1.f, nullptr, 0, 0.f, 0, nullptr, nin);
        if (rc) return rc;
        return vlt5_glu_fwd(u, h, M, ff, dp, dseed, st);
    }
    // y = resid + dropout(x W^T) for a sublayer output that only a LayerNorm consumes next: when the reduction is long and the output
    // small (decoder FFN: 400 x 768 over K = 3072, 84 tiles for 256 CUs) the GEMM is cut into split-K slices that stay in the slab
    // scratch, and ln_fwd_pending() assembles the row (slab sum, dropout, residual) while it normalises it.  *pending = number
    // of slabs left for the norm (0: y was written by the GEMM's own epilogue as usual).
    int lin_fwd_for_norm(const bf16_t* X, const bf16_t* W, float* Y, int M, int N, int K, float dp, uint32_t dseed,
                         const float* resid, int* pending) const {
        *pending = 0;
        const int sk = pick_split(M, N, K);
        if (sk <= 1) return lin_fwd(X, W, Y, M, N, K, 1, 1.f, nullptr, 0, dp, dseed, resid);
        vlt5_gemm_desc g;
        memset(&g, 0, sizeof g);
        g.tuning = &tun;
        g.A = X; g.B = W; g.C = w<void>(p.slab); g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N;
        g.alpha = 1.f; g.out_f32 = 1; g.split_k = sk; g.workspace = w<void>(p.slab); g.defer_reduce = 1;
        int rc = vlt5_gemm_bf16(&g, st);
        *pending = g.split_used;             // (1: the policy clipped the split; the un-split product lies in the slab scratch too)
        return rc;
    }
    // LayerNorm forward of x, or -- pending > 0 -- of resid + dropout(sum of the pending slabs), which is also stored to x
    int ln_fwd_pending(int pending, float* x, const float* resid, float rdp, uint32_t rseed, long long w_off, void* y_bf16, float* y_f32,
                       float* rstd, int rows, float dp, uint32_t dseed, int og, int ogs) const {
        if (pending > 0)
            return vlt5_layernorm_fwd_slabs(w<float>(p.slab), pending, (long long)rows * d, resid, x, rdp, rseed, P + w_off, y_bf16, y_f32,
                                            rstd, rows, d, c.eps, dp, dseed, og, ogs, st);
        return vlt5_layernorm_fwd(x, P + w_off, y_bf16, y_f32, rstd, rows, d, c.eps, dp, dseed, og, ogs, st);
    }
    // dX[M,K] = epi(alpha * dY[M,N] W[N,K])     (W read k-major)
    // `slabs` (optional): the consumer is a LayerNorm backward that can sum split-K slabs itself -- on return *slabs = number of
    // slabs left in the slab scratch (1: dX was written as usual)
    int lin_dgrad(const bf16_t* dY, const bf16_t* W, void* dX, int M, int N, int K, int out_f32, float alpha = 1.f,
                  const bf16_t* gate = nullptr, float gate_scale = 1.f, int* slabs = nullptr, const void* gate_bits = nullptr) const {
        vlt5_gemm_desc g;
        memset(&g, 0, sizeof g);
        g.tuning = &tun;
        g.A = dY; g.B = W; g.C = dX; g.M = M; g.N = K; g.K = N; g.lda = N; g.ldb = K; g.ldc = K; g.b_kmajor = 1;
        g.alpha = alpha; g.gate = gate; g.ldg = K; g.gate_scale = gate_scale; g.out_f32 = out_f32;
        if (gate_bits) { g.gate = nullptr; g.gate_bits = gate_bits; g.ld_bits = K / 8; }
        if (out_f32 && !gate && !gate_bits) {
            int sk = pick_split(M, K, N);
            if (sk > 1) { g.split_k = sk; g.workspace = w<void>(p.slab); g.defer_reduce = slabs ? 1 : 0; }
        }
        int rc = vlt5_gemm_bf16(&g, st);
        if (slabs) *slabs = g.split_used > 1 ? g.split_used : 1;
        return rc;
    }
    // dW[N,K] (+)= alpha * dY[M,N]^T X[M,K]     (both read k-major, reduction over the M rows)
    // norm_share: the tiles' sums of squares go to the gradient-norm partials (a tensor nothing else adds to afterwards)
    int lin_wgrad(const bf16_t* dY, int ldy, const bf16_t* X, int ldx, float* dW, int M, int N, int K, float alpha = 1.f,
                  int accum = 0, bool norm_share = false, const vlt5_gemm_desc* group = nullptr) const {
        vlt5_gemm_desc g;
        memset(&g, 0, sizeof g);
        g.tuning = &tun;
        g.grouped_with = group;
        g.A = dY; g.B = X; g.C = dW; g.M = N; g.N = K; g.K = M; g.lda = ldy; g.ldb = ldx; g.ldc = K;
        g.a_kmajor = 1; g.b_kmajor = 1; g.alpha = alpha; g.out_f32 = 1; g.accum = accum;
        if (!accum) g.c_bf16_copy = g16(dW);
        if (norm_share && !accum) g.sumsq = gsq(dW);
        int sk = (g.sumsq || group) ? 1 : pick_split(N, K, M);   // (the slab reduction has no norm share: such launches are not cut)
        if (sk > 1) { g.split_k = sk; g.workspace = w<void>(p.slab); }
        return vlt5_gemm_bf16(&g, st);
    }
    // slot of a gradient tensor in the gradient-norm partials (vlt5_step.gnorm_partials): ceil(flat offset / 4096); tensors whose
    // size is a multiple of 4096 elements (both dimensions multiples of 64) then own disjoint slot ranges of size / 4096
    bool gnorm_ok() const { return s.gnorm_partials && !(d & 63) && !(inner & 63) && !(ffw() & 63); }
    float* gsq(const float* dW) const { return gnorm_ok() ? s.gnorm_partials + ((dW - Gr) + 4095) / 4096 : nullptr; }
    // the bf16 mirror of a gradient tensor (vlt5_step.grads_bf16), or null
    void* g16(const float* dW) const {
        return s.grads_bf16 ? (void*)((bf16_t*)s.grads_bf16 + (dW - Gr)) : nullptr;
    }
    int mirror_small(long long off, long long n) const {       // gradients no GEMM writes (relative-position tables): cast them
        if (!s.grads_bf16) return VLT5_OK;
        return vlt5_cast_bf16(Gr + off, (bf16_t*)s.grads_bf16 + off, n, st);
    }
    // T5LayerNorm backward into the running residual gradient `dx`; `emit_next` additionally writes bf16(dropout(dx)) into
    // `next_dst` (the per-layer operand buffer of the sublayer processed next).  The weight gradient is left as per-workgroup partials in
    // slot `ln_jobs`; ln_flush() reduces all slots of the phase with one launch.
    int ln_bwd(const float* dy, const float* x, long long w_off, const float* rstd, float* dx, int rows, int accum_dx,
               float dp, uint32_t dseed, int in_group, int in_group_stride, bf16_t* next_dst, uint32_t next_seed,
               int nslabs = 1, long long slab_stride = 0, bf16_t* xn_out = nullptr) const {
        if (ln_jobs >= 64) { int rc = ln_flush(); if (rc) return rc; }     // deep stacks (t5-large: 73 norms per phase)
        float* part = w<float>(p.ln_partial) + (size_t)ln_jobs * LNB_MAXBLK * d;
        ln_out[ln_jobs] = w_off;
        ln_nblk[ln_jobs] = vlt5_layernorm_bwd_blocks(rows);
        ++ln_jobs;
        if (nslabs > 1) dy = w<float>(p.slab);            // the producing GEMM left its split-K slabs there
        // xn_out: the norm was folded around its GEMMs in the forward -- its bf16 output, the operand of the weight gradient of the
        // projection behind it, is written now
        return vlt5_layernorm_bwd_full(dy, nslabs, slab_stride, x, P + w_off, rstd, dx, nullptr, part, rows, d, accum_dx, 0, dp,
                                       dseed, in_group, in_group_stride, next_dst, next_dst ? pdrop : 0.f, next_seed, xn_out, st);
    }
    int ln_flush() const {
        if (ln_jobs == 0) return VLT5_OK;
        int rc = vlt5_colsum_multi(w<float>(p.ln_partial), Gr, ln_out, ln_nblk, ln_jobs, LNB_MAXBLK, d, st);
        ln_jobs = 0;
        return rc;
    }
    // dW_l[N,K] = dY_l[M,N]^T X_l[M,K] for `layers` layers at once; per-layer byte strides of the operand buffers and the
    // element stride of the gradient entries are constant by construction of the plan / layout.
    int wgrad_batched(size_t dy0, size_t dy1, int ldy, size_t x0, size_t x1, int ldx, long long g0, long long g1, int layers,
                      int M, int N, int K, vlt5_gemm_desc* only_fill = nullptr, const vlt5_gemm_desc* group = nullptr) const {
        vlt5_gemm_desc g;
        memset(&g, 0, sizeof g);
        g.tuning = &tun;
        g.grouped_with = group;
        g.A = w<void>(dy0); g.B = w<void>(x0); g.C = Gr + g0; g.M = N; g.N = K; g.K = M; g.lda = ldy; g.ldb = ldx; g.ldc = K;
        g.a_kmajor = 1; g.b_kmajor = 1; g.alpha = 1.f; g.out_f32 = 1;
        g.c_bf16_copy = g16(Gr + g0);
        g.batch = layers;
        g.batch_stride_a = layers > 1 ? ((long long)dy1 - (long long)dy0) / 2 : 0;
        g.batch_stride_b = layers > 1 ? ((long long)x1 - (long long)x0) / 2 : 0;
        g.batch_stride_c = layers > 1 ? (g1 - g0) : 0;
        g.sumsq = gsq(Gr + g0);                                // every batched weight gradient is a layer matrix: norm share
        g.sumsq_batch_stride = layers > 1 ? (g1 - g0) / 4096 : 0;
        if (layers == 1 && !g.sumsq && !group && !only_fill) {
            int sk = pick_split(N, K, M);
            if (sk > 1) { g.split_k = sk; g.workspace = w<void>(p.slab); }
        }
        if (only_fill) { *only_fill = g; return VLT5_OK; }       // (the second problem of a grouped launch)
        return vlt5_gemm_bf16(&g, st);
    }
    // forward side of an overlapped optimizer: wait until parameter bucket b has been updated (no-op without events)
    int wait_bucket(int b) const {
        if (s.wait_events && b >= 0 && b < s.n_wait_events && s.wait_events[b] && s.wait_events[b] != last_wait) {
            // (consecutive buckets may share one event -- the all-gather of a merged slice under sharded data parallelism: one wait
            // per slice; every cross-stream wait is a barrier packet in the chain's queue)
            HIP_RET(hipStreamWaitEvent(st, (hipEvent_t)s.wait_events[b], 0));
            last_wait = s.wait_events[b];
        }
        return VLT5_OK;
    }
    int record(int k) const {
        if (s.events && k >= 0 && k < s.n_events && s.events[k]) HIP_RET(hipEventRecord((hipEvent_t)s.events[k], st));
        return VLT5_OK;
    }
};

#define RC(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

int dec_wgrad(const Ctx& k, int which, vlt5_gemm_desc* desc, int lo = 0, int n = 0, const vlt5_gemm_desc* group = nullptr);
bool shadow_wgrads(const Ctx& k);
bool bucket_events(const Ctx& k);
bool shadow_rule(const vlt5_config& c, const vlt5_tuning* t, bool events, bool side);

int attn_call(const Ctx& k, bool bwd, const bf16_t* q, long long q_sb, long long q_st, const bf16_t* kk, const bf16_t* v,
              long long kv_sb, long long kv_st, bf16_t* ctx, float* lse, const float* bias, int bq, int bk, const float* kmask,
              float mval, int causal, int Tq, int Tk, uint32_t dseed, const bf16_t* dctx = nullptr, bf16_t* dq = nullptr,
              long long dq_sb = 0, long long dq_st = 0, bf16_t* dk = nullptr, bf16_t* dv = nullptr, long long dkv_sb = 0,
              long long dkv_st = 0, float* dbias = nullptr) {
    vlt5_attn_desc a;
    memset(&a, 0, sizeof a);
    a.q = q; a.k = kk; a.v = v; a.q_sb = q_sb; a.q_st = q_st; a.k_sb = kv_sb; a.k_st = kv_st; a.v_sb = kv_sb; a.v_st = kv_st;
    a.ctx = ctx; a.o_sb = (long long)Tq * k.inner; a.o_st = k.inner; a.lse = lse;
    a.bias = bias; a.bias_q = bq; a.bias_k = bk; a.key_mask = kmask; a.mask_value = mval; a.causal = causal;
    a.B = k.s.B; a.H = k.H; a.Tq = Tq; a.Tk = Tk; a.dk = k.c.d_kv; a.drop_p = k.pdrop; a.drop_seed = dseed;
    if (!bwd) return vlt5_attn_fwd(&a, k.st);
    a.d_ctx = dctx; a.do_sb = (long long)Tq * k.inner; a.do_st = k.inner;
    a.dq = dq; a.dq_sb = dq_sb; a.dq_st = dq_st; a.dk_ = dk; a.dv = dv; a.dk_sb = dkv_sb; a.dk_st = dkv_st; a.dv_sb = dkv_sb;
    a.dv_st = dkv_st; a.dbias = dbias;
    return vlt5_attn_bwd(&a, k.st);
}

// the fused q|k|v projection + attention core kernel (csrc/enc_attn.hip) covers d_kv = 64, S <= 64 and
// d_model % 64 == 0 (every T5 size); vlt5_tuning.fused_attn = 1 selects the GEMM + attention-core launches instead (A/B runs, tests)
// By default only where its grid fills the chip: a workgroup is 2 samples x 2 heads, B / 2 x H / 2 of them -- below ~128 a launch takes the
// 26 us of ONE workgroup whatever B is, and the separate projection + core launches win (B = 4 ... 32 at 12 heads: 4.86 / 5.19 / 5.64 /
// 6.57 ms per step against 5.01 / 5.31 / 5.80 / 6.62; B = 48, 144 workgroups: 7.16 against 7.19).  fused_attn = 2 forces it.
bool fused_attn_ok(const Ctx& k) {
    if (k.tun.fused_attn == 1 || k.c.d_kv != 64 || k.p.S > 64 || (k.d & 63) != 0) return false;
    return k.tun.fused_attn == 2 || ((k.s.B + 1) / 2) * (k.H / 2) >= 128;
}

// the fused decoder attention sublayers (csrc/dec_attn.hip: projection + core + per-head output-projection slabs in one launch instead of
// three): built and bit-checked, measured level with the three launches they replace -- vlt5_tuning.dec_fused = 2 selects them
bool dec_fused_ok(const Ctx& k) {
    return k.tun.dec_fused == 2 && !k.fold_dec() && vlt5_dec_attn_fused_ok(k.s.T, k.p.Sx, k.c.d_kv, k.d) &&
           (size_t)k.H * k.p.Md * k.d * sizeof(float) <= k.p.slab_bytes;
}
// one fused decoder attention sublayer: leaves H slabs in the slab scratch for the norm that follows
int dec_attn_fused(const Ctx& k, bool cross, const bf16_t* xn, const bf16_t* w, const bf16_t* wo, bf16_t* proj, const bf16_t* kk,
                   const bf16_t* v, long long kv_sb, long long kv_st, bf16_t* ctx, float* lse, const float* bias, const float* kmask,
                   float mval, int Tk, uint32_t dseed) {
    const int T = k.s.T, inner = k.inner;
    vlt5_dec_attn_desc e;
    memset(&e, 0, sizeof e);
    e.xn_bf16 = xn; e.w_bf16 = w; e.wo_bf16 = wo; e.proj_bf16 = proj; e.o_slabs = k.w<float>(k.p.slab);
    e.slab_stride = (long long)k.p.Md * k.d; e.d_model = k.d;
    vlt5_attn_desc& a = e.core;
    if (cross) {
        a.q = proj; a.q_sb = (long long)T * inner; a.q_st = inner; a.k = kk; a.v = v; a.k_sb = a.v_sb = kv_sb; a.k_st = a.v_st = kv_st;
    } else {
        a.q = proj; a.k = proj + inner; a.v = proj + 2 * inner;
        a.q_sb = a.k_sb = a.v_sb = (long long)T * 3 * inner; a.q_st = a.k_st = a.v_st = 3 * inner;
    }
    a.ctx = ctx; a.o_sb = (long long)T * inner; a.o_st = inner; a.lse = lse;
    a.bias = bias; a.bias_q = bias ? T : 0; a.bias_k = bias ? T : 0; a.key_mask = kmask; a.mask_value = mval; a.causal = cross ? 0 : 1;
    a.B = k.s.B; a.H = k.H; a.Tq = T; a.Tk = Tk; a.dk = 64; a.drop_p = k.pdrop; a.drop_seed = dseed;
    return cross ? vlt5_cross_attn_fwd(&e, k.st) : vlt5_dec_self_attn_fwd(&e, k.st);
}

int encoder_fwd(const Ctx& k) {
    const Plan& p = k.p; const Layout& L = k.lay; const vlt5_config& c = k.c; const vlt5_step& s = k.s;
    const int d = k.d, inner = k.inner, ff = k.ff, M = p.M, S = p.S, Sx = p.Sx, B = s.B;
    float* x0 = k.w<float>(p.x[0]);
    const int nb = c.num_decoder_layers + c.num_layers + 2;      // buckets: decoder top..0, cross k/v, encoder top..0, norms+embeddings
    RC(k.wait_bucket(nb - 1));
    {   // key mask, relative-position bias block and the text rows of the input embeddings: one launch
        vlt5_stack_inputs_desc si;
        memset(&si, 0, sizeof si);
        si.mask_ids = s.input_ids; si.mask = k.w<float>(p.mask); si.B = B; si.L = s.L; si.S = S;
        si.rel_table = k.P + L.enc_rel; si.lut = s.enc_lut; si.bias = k.w<float>(p.enc_bias); si.H = k.H; si.Lq = s.L; si.Lk = s.L;
        si.ids = s.input_ids; si.T = s.L; si.pad_id = c.pad_id;
        si.table = k.P + L.shared; si.out = x0; si.out_sb = (long long)S * d; si.out_st = d; si.d = d; si.vocab = c.vocab;
        si.drop_p = k.pdrop; si.drop_seed = k.seed(SITE_ENC_EMBED); si.drop_rows = S; si.drop_row0 = 0;
        RC(vlt5_stack_inputs_fwd(&si, k.st));
    }
    if (s.feat_store)       // batch assembled from the resident store: bf16 rows as they are (the rounding the cast would apply)
        RC(vlt5_feat_gather(s.feat_store, s.box_store, s.feat_slots, s.n_slots, k.w<void>(p.feats_bf16), k.w<float>(p.boxes_g), B, s.V,
                            c.feat_dim, k.st));
    else
        RC(vlt5_cast_bf16(s.vis_feats, k.w<void>(p.feats_bf16), (long long)B * s.V * c.feat_dim, k.st));
    RC(k.lin_fwd(k.w<bf16_t>(p.feats_bf16), k.Pb + L.vis_wf, k.w<void>(p.visG), B * s.V, d, c.feat_dim, 1, 1.f, k.P + L.vis_bf));
    RC(vlt5_vis_embed_fwd(k.w<float>(p.visG), k.boxes(), k.P + L.vis_wp, k.P + L.vis_bp, k.P + L.vis_lnf, k.P + L.vis_lnp,
                          k.P + L.vis_img, k.P + L.shared, x0 + (size_t)s.L * d, (long long)S * d, d, k.w<float>(p.vis_rf),
                          k.w<float>(p.vis_rp), B, s.V, d, c.vocab, c.eps, k.pdrop, k.seed(SITE_ENC_EMBED), S, s.L, k.st));
    int pending = 0;                                          // split-K slabs of the previous layer's FFN output, if any
    int np_a = 0;                                             // > 0: the norm in front of this layer's attention is folded (partials per row)
    for (int l = 0; l < c.num_layers; ++l) {
        const auto& E = L.enc[l];
        const uint32_t sb = SITE_ENC_BASE + l * 8;
        RC(k.wait_bucket(c.num_decoder_layers + c.num_layers - l));
        float* xa = k.w<float>(p.x[2 * l]);
        float* xf = k.w<float>(p.x[2 * l + 1]);
        float* xo = k.w<float>(p.x[2 * l + 2]);
        bf16_t* qkv = k.w<bf16_t>(p.qkv[l]);
        // T5 RMS norm + q|k|v projection + attention core.  Folded norm (np_a > 0): the FFN output GEMM of the previous layer left
        // bf16(x * w) in xn_a and the rows' partial sums of squares; the projection's rows are scaled by rstd, no norm launch
        if (np_a == 0)
            RC(k.ln_fwd_pending(pending, xa, l > 0 ? k.w<float>(p.x[2 * l - 1]) : nullptr, k.pdrop, k.seed(sb - 8 + E_FFN_OUT), E.ln_s,
                                k.w<void>(p.xn_a[l]), nullptr, k.w<float>(p.xr[2 * l]), M, 0.f, 0, 0, 0));
        NormIn nin_a;
        if (np_a > 0) { nin_a.part = k.w<float>(p.ssq_e[2 * l]); nin_a.n = np_a; nin_a.rstd_out = k.w<float>(p.xr[2 * l]); }
        if (fused_attn_ok(k) && np_a <= 32) {
            vlt5_attn_desc a;
            memset(&a, 0, sizeof a);
            a.q = qkv; a.k = qkv + inner; a.v = qkv + 2 * inner;
            a.q_sb = a.k_sb = a.v_sb = (long long)S * 3 * inner; a.q_st = a.k_st = a.v_st = 3 * inner;
            a.ctx = k.w<void>(p.ctx[l]); a.o_sb = (long long)S * inner; a.o_st = inner; a.lse = k.w<float>(p.lse[l]);
            a.bias = k.w<float>(p.enc_bias); a.bias_q = s.L; a.bias_k = s.L; a.key_mask = k.w<float>(p.mask); a.mask_value = -10000.f;
            a.B = B; a.H = k.H; a.Tq = S; a.Tk = S; a.dk = c.d_kv; a.drop_p = k.pdrop; a.drop_seed = k.seed(sb + E_PROBS);
            a.fused_heads = k.tun.fused_heads;
            if (np_a > 0)
                RC(vlt5_qkv_attn_fwd_norm(k.w<void>(p.xn_a[l]), k.Pb + E.sqkv, qkv, &a, d, nin_a.part, nin_a.n, c.eps, nin_a.rstd_out, k.st));
            else
                RC(vlt5_qkv_attn_fwd(k.w<void>(p.xn_a[l]), k.Pb + E.sqkv, qkv, &a, d, k.st));
        } else {
            RC(k.lin_fwd(k.w<bf16_t>(p.xn_a[l]), k.Pb + E.sqkv, qkv, M, 3 * inner, d, 0, 1.f, nullptr, 0, 0.f, 0, nullptr, nin_a));
            RC(attn_call(k, false, qkv, (long long)S * 3 * inner, 3 * inner, qkv + inner, qkv + 2 * inner, (long long)S * 3 * inner,
                         3 * inner, k.w<bf16_t>(p.ctx[l]), k.w<float>(p.lse[l]), k.w<float>(p.enc_bias), s.L, s.L, k.w<float>(p.mask),
                         -10000.f, 0, S, S, k.seed(sb + E_PROBS)));
        }
        // output projection (+ dropout + residual); folded: its epilogue also leaves the operand and the partials of the FFN's norm
        NormIn nin_f;
        if (k.fold_on()) {
            int np_f = 0;
            RC(k.lin_fwd_emit(k.w<bf16_t>(p.ctx[l]), k.Pb + E.so, xf, M, d, inner, k.pdrop, k.seed(sb + E_ATTN_OUT), xa, E.ln_f,
                              k.w<void>(p.xn_f[l]), k.w<float>(p.ssq_e[2 * l + 1]), &np_f, true));
            nin_f.part = k.w<float>(p.ssq_e[2 * l + 1]); nin_f.n = np_f; nin_f.rstd_out = k.w<float>(p.xr[2 * l + 1]);
        } else {
            RC(k.lin_fwd(k.w<bf16_t>(p.ctx[l]), k.Pb + E.so, xf, M, d, inner, 1, 1.f, nullptr, 0, k.pdrop, k.seed(sb + E_ATTN_OUT), xa));
            RC(vlt5_layernorm_fwd(xf, k.P + E.ln_f, k.w<void>(p.xn_f[l]), nullptr, k.w<float>(p.xr[2 * l + 1]), M, d, c.eps, 0.f, 0, 0, 0, k.st));
        }
        RC(k.ffn_hidden(k.w<bf16_t>(p.xn_f[l]), E.wi, k.w<bf16_t>(p.u[l]), k.w<bf16_t>(p.h[l]), M, k.pdrop, k.seed(sb + E_FFN_H), nin_f,
                        k.bits_on() ? k.w<void>(p.hbits[l]) : nullptr));
        np_a = 0;
        if (l + 1 < c.num_layers && k.fold_enc_first(l + 1)) {
            pending = 0;
            RC(k.lin_fwd_emit(k.w<bf16_t>(p.h[l]), k.Pb + E.wo, xo, M, d, ff, k.pdrop, k.seed(sb + E_FFN_OUT), xf, L.enc[l + 1].ln_s,
                              k.w<void>(p.xn_a[l + 1]), k.w<float>(p.ssq_e[2 * l + 2]), &np_a, true));
        } else {
            RC(k.lin_fwd_for_norm(k.w<bf16_t>(p.h[l]), k.Pb + E.wo, xo, M, d, ff, k.pdrop, k.seed(sb + E_FFN_OUT), xf, &pending));
        }
    }
    const int Le = c.num_layers;
    RC(k.ln_fwd_pending(pending, k.w<float>(p.x[2 * Le]), k.w<float>(p.x[2 * Le - 1]), k.pdrop,
                        k.seed(SITE_ENC_BASE + (Le - 1) * 8 + E_FFN_OUT), L.enc_final_ln, k.w<void>(p.enc_ext), k.w<float>(p.enc_out),
                        k.w<float>(p.xr[2 * Le]), M, k.pdrop, k.seed(SITE_ENC_FINAL), S, Sx));
    return VLT5_OK;
}

int decoder_fwd(const Ctx& k) {
    const Plan& p = k.p; const Layout& L = k.lay; const vlt5_config& c = k.c; const vlt5_step& s = k.s;
    const int d = k.d, inner = k.inner, ff = k.ff, Md = p.Md, Mx = p.Mx, Sx = p.Sx, B = s.B, T = s.T, Ld = c.num_decoder_layers;
    const int kvw = Ld * 2 * inner;
    long long* ids = k.w<long long>(p.dec_ids);
    RC(k.wait_bucket(Ld + c.num_layers + 1));                 // (a decoder-only call: norms + embeddings first)
    RC(k.wait_bucket(Ld));                                    // stacked cross-attention k/v
    {   // shift-right of the labels, extended key mask, causal relative-position bias block, decoder input embeddings: one launch
        vlt5_stack_inputs_desc si;
        memset(&si, 0, sizeof si);
        si.mask_ids = s.input_ids; si.mask = k.w<float>(p.mask_ext); si.B = B; si.L = s.L; si.S = Sx;
        si.rel_table = k.P + L.dec_rel; si.lut = s.dec_lut; si.bias = k.w<float>(p.dec_bias); si.H = k.H; si.Lq = T; si.Lk = T;
        si.labels = s.labels; si.ids_out = ids; si.T = T; si.start_id = c.dec_start_id; si.pad_id = c.pad_id;
        si.table = k.P + L.shared; si.out = k.w<float>(p.y[0]); si.out_sb = (long long)T * d; si.out_st = d; si.d = d; si.vocab = c.vocab;
        si.drop_p = k.pdrop; si.drop_seed = k.seed(SITE_DEC_EMBED); si.drop_rows = T; si.drop_row0 = 0;
        RC(vlt5_stack_inputs_fwd(&si, k.st));
    }
    RC(k.lin_fwd(k.w<bf16_t>(p.enc_ext), k.Pb + L.cross_kv, k.w<void>(p.kv_all), Mx, kvw, d, 0));
    int pending = 0;                                          // split-K slabs of the previous layer's FFN output, if any
    for (int l = 0; l < Ld; ++l) {
        const auto& D = L.dec[l];
        const uint32_t sb = SITE_DEC_BASE + l * 8;
        RC(k.wait_bucket(Ld - 1 - l));
        float* y0 = k.w<float>(p.y[3 * l]);
        float* y1 = k.w<float>(p.y[3 * l + 1]);
        float* y2 = k.w<float>(p.y[3 * l + 2]);
        float* y3 = k.w<float>(p.y[3 * l + 3]);
        bf16_t* qkv = k.w<bf16_t>(p.dqkv_s[l]);
        bf16_t* kv = k.w<bf16_t>(p.kv_all) + (size_t)l * 2 * inner;
        RC(k.ln_fwd_pending(pending, y0, l > 0 ? k.w<float>(p.y[3 * l - 1]) : nullptr, k.pdrop, k.seed(sb - 8 + D_FFN_OUT), D.ln_s,
                            k.w<void>(p.yn_a[l]), nullptr, k.w<float>(p.yr[3 * l]), Md, 0.f, 0, 0, 0));
        if (dec_fused_ok(k)) {
            // three launches per attention sublayer in one: the H slabs of each output projection are summed by the norm behind it
            RC(dec_attn_fused(k, false, k.w<bf16_t>(p.yn_a[l]), k.Pb + D.sqkv, k.Pb + D.so, qkv, nullptr, nullptr, 0, 0, k.w<bf16_t>(p.ctx_s[l]),
                              k.w<float>(p.lse_s[l]), k.w<float>(p.dec_bias), nullptr, 0.f, T, k.seed(sb + D_SPROBS)));
            RC(k.ln_fwd_pending(k.H, y1, y0, k.pdrop, k.seed(sb + D_SOUT), D.ln_c, k.w<void>(p.yn_c[l]), nullptr,
                                k.w<float>(p.yr[3 * l + 1]), Md, 0.f, 0, 0, 0));
            RC(dec_attn_fused(k, true, k.w<bf16_t>(p.yn_c[l]), k.Pb + D.cq, k.Pb + D.co, k.w<bf16_t>(p.qc[l]), kv, kv + inner,
                              (long long)Sx * kvw, kvw, k.w<bf16_t>(p.ctx_c[l]), k.w<float>(p.lse_c[l]), nullptr, k.w<float>(p.mask_ext), -1e9f,
                              Sx, k.seed(sb + D_CPROBS)));
            RC(k.ln_fwd_pending(k.H, y2, y1, k.pdrop, k.seed(sb + D_COUT), D.ln_f, k.w<void>(p.yn_f[l]), nullptr,
                                k.w<float>(p.yr[3 * l + 2]), Md, 0.f, 0, 0, 0));
            RC(k.ffn_hidden(k.w<bf16_t>(p.yn_f[l]), D.wi, k.w<bf16_t>(p.ud[l]), k.w<bf16_t>(p.hd[l]), Md, k.pdrop, k.seed(sb + D_FFN_H)));
            RC(k.lin_fwd_for_norm(k.w<bf16_t>(p.hd[l]), k.Pb + D.wo, y3, Md, d, ff, k.pdrop, k.seed(sb + D_FFN_OUT), y2, &pending));
            continue;
        }
        RC(k.lin_fwd(k.w<bf16_t>(p.yn_a[l]), k.Pb + D.sqkv, qkv, Md, 3 * inner, d, 0));
        RC(attn_call(k, false, qkv, (long long)T * 3 * inner, 3 * inner, qkv + inner, qkv + 2 * inner, (long long)T * 3 * inner,
                     3 * inner, k.w<bf16_t>(p.ctx_s[l]), k.w<float>(p.lse_s[l]), k.w<float>(p.dec_bias), T, T, nullptr, 0.f, 1, T, T,
                     k.seed(sb + D_SPROBS)));
        // (the two attention output projections, 84 tiles over K = 768, are cut along K like the FFN output: the norm that follows
        // sums the slabs while it assembles the row)
        if (k.fold_dec()) {
            // folded norms: the two attention output projections run un-split and leave bf16(y * w) + the rows' partial sums of
            // squares for the norm that follows; the cross-attention query projection / the FFN input projection scale their rows
            // by rstd -- two norm launches less per layer on the decoder's latency chain
            int np_c = 0, np_f = 0;
            RC(k.lin_fwd_emit(k.w<bf16_t>(p.ctx_s[l]), k.Pb + D.so, y1, Md, d, inner, k.pdrop, k.seed(sb + D_SOUT), y0, D.ln_c,
                              k.w<void>(p.yn_c[l]), k.w<float>(p.ssq_d[3 * l + 1]), &np_c));
            NormIn nc; nc.part = k.w<float>(p.ssq_d[3 * l + 1]); nc.n = np_c; nc.rstd_out = k.w<float>(p.yr[3 * l + 1]);
            RC(k.lin_fwd(k.w<bf16_t>(p.yn_c[l]), k.Pb + D.cq, k.w<void>(p.qc[l]), Md, inner, d, 0, 1.f, nullptr, 0, 0.f, 0, nullptr, nc));
            RC(attn_call(k, false, k.w<bf16_t>(p.qc[l]), (long long)T * inner, inner, kv, kv + inner, (long long)Sx * kvw, kvw,
                         k.w<bf16_t>(p.ctx_c[l]), k.w<float>(p.lse_c[l]), nullptr, 0, 0, k.w<float>(p.mask_ext), -1e9f, 0, T, Sx,
                         k.seed(sb + D_CPROBS)));
            RC(k.lin_fwd_emit(k.w<bf16_t>(p.ctx_c[l]), k.Pb + D.co, y2, Md, d, inner, k.pdrop, k.seed(sb + D_COUT), y1, D.ln_f,
                              k.w<void>(p.yn_f[l]), k.w<float>(p.ssq_d[3 * l + 2]), &np_f));
            NormIn nf; nf.part = k.w<float>(p.ssq_d[3 * l + 2]); nf.n = np_f; nf.rstd_out = k.w<float>(p.yr[3 * l + 2]);
            RC(k.ffn_hidden(k.w<bf16_t>(p.yn_f[l]), D.wi, k.w<bf16_t>(p.ud[l]), k.w<bf16_t>(p.hd[l]), Md, k.pdrop, k.seed(sb + D_FFN_H), nf));
        } else {
        int pend_s = 0, pend_c = 0;
        RC(k.lin_fwd_for_norm(k.w<bf16_t>(p.ctx_s[l]), k.Pb + D.so, y1, Md, d, inner, k.pdrop, k.seed(sb + D_SOUT), y0, &pend_s));
        RC(k.ln_fwd_pending(pend_s, y1, y0, k.pdrop, k.seed(sb + D_SOUT), D.ln_c, k.w<void>(p.yn_c[l]), nullptr,
                            k.w<float>(p.yr[3 * l + 1]), Md, 0.f, 0, 0, 0));
        RC(k.lin_fwd(k.w<bf16_t>(p.yn_c[l]), k.Pb + D.cq, k.w<void>(p.qc[l]), Md, inner, d, 0));
        RC(attn_call(k, false, k.w<bf16_t>(p.qc[l]), (long long)T * inner, inner, kv, kv + inner, (long long)Sx * kvw, kvw,
                     k.w<bf16_t>(p.ctx_c[l]), k.w<float>(p.lse_c[l]), nullptr, 0, 0, k.w<float>(p.mask_ext), -1e9f, 0, T, Sx,
                     k.seed(sb + D_CPROBS)));
        RC(k.lin_fwd_for_norm(k.w<bf16_t>(p.ctx_c[l]), k.Pb + D.co, y2, Md, d, inner, k.pdrop, k.seed(sb + D_COUT), y1, &pend_c));
        RC(k.ln_fwd_pending(pend_c, y2, y1, k.pdrop, k.seed(sb + D_COUT), D.ln_f, k.w<void>(p.yn_f[l]), nullptr,
                            k.w<float>(p.yr[3 * l + 2]), Md, 0.f, 0, 0, 0));
        RC(k.ffn_hidden(k.w<bf16_t>(p.yn_f[l]), D.wi, k.w<bf16_t>(p.ud[l]), k.w<bf16_t>(p.hd[l]), Md, k.pdrop, k.seed(sb + D_FFN_H)));
        }
        RC(k.lin_fwd_for_norm(k.w<bf16_t>(p.hd[l]), k.Pb + D.wo, y3, Md, d, ff, k.pdrop, k.seed(sb + D_FFN_OUT), y2, &pending));
    }
    RC(k.ln_fwd_pending(pending, k.w<float>(p.y[3 * Ld]), k.w<float>(p.y[3 * Ld - 1]), k.pdrop,
                        k.seed(SITE_DEC_BASE + (Ld - 1) * 8 + D_FFN_OUT), L.dec_final_ln, k.w<void>(p.dec_out), nullptr,
                        k.w<float>(p.yr[3 * Ld]), Md, k.pdrop, k.seed(SITE_DEC_FINAL), 0, 0));
    const float alpha = 1.0f / sqrtf((float)d);           // tied embeddings: rescale before the vocabulary projection
    RC(k.lin_fwd(k.w<bf16_t>(p.dec_out), k.Pb + L.shared, k.w<void>(p.logits), Md, c.vocab, d, 1, alpha));
    RC(vlt5_ce_fwd(k.w<float>(p.logits), s.labels, k.w<float>(p.loss_tok), k.w<float>(p.lse_ce), Md, c.vocab, k.st));
    if (s.scores)
        RC(vlt5_loss_reduce(k.w<float>(p.loss_tok), s.labels, s.scores, k.w<float>(p.loss), k.w<float>(p.row_w), B, T, k.st));
    return VLT5_OK;
}

// One greedy-decoding step with a key/value cache (reference: HF generate -> VLT5.forward(decoder_input_ids[:, -1:],
// past_key_values), src/modeling_t5_our.py:544-566,624-629; vqa_model.py:112-116).  Row b of `tokens` is the decoder input at
// position t; the self-attention keys/values of positions 0..t live in `cache` ([Ld][B][Tcap][2*inner] bf16, Tcap = s.T), the
// cross-attention keys/values of the 58 encoder rows are computed at t == 0 (the encoder output incl. the two retrieved
// prototype rows must be in the workspace: vlt5_encoder_fwd + prototype retrieval of the same step state).  Eval mode only.
int decoder_step(const Ctx& k, const long long* tokens, int t, bf16_t* cache, float* logits, long long* next_ids) {
    const Plan& p = k.p; const Layout& L = k.lay; const vlt5_config& c = k.c; const vlt5_step& s = k.s;
    const int d = k.d, inner = k.inner, ff = k.ff, Mx = p.Mx, Sx = p.Sx, B = s.B, Tcap = s.T, Ld = c.num_decoder_layers;
    const int kvw = Ld * 2 * inner, Tk = t + 1;
    if (t == 0) {
        RC(k.wait_bucket(Ld + c.num_layers + 1));
        RC(k.wait_bucket(Ld));
        RC(vlt5_build_mask(s.input_ids, k.w<float>(p.mask_ext), B, s.L, Sx, c.pad_id, k.st));
        RC(k.lin_fwd(k.w<bf16_t>(p.enc_ext), k.Pb + L.cross_kv, k.w<void>(p.kv_all), Mx, kvw, d, 0));
        for (int l = 0; l < Ld; ++l) RC(k.wait_bucket(Ld - 1 - l));
    }
    // relative-position bias of query position t against keys 0..t: row t of the causal bucket table, [H][1][Tk]
    float* bias_t = k.w<float>(p.dec_bias);
    RC(vlt5_relbias_build(k.P + L.dec_rel, s.dec_lut + (size_t)t * Tcap, bias_t, k.H, 1, Tk, c.rel_buckets, k.st));
    RC(vlt5_embed_fwd(tokens, k.P + L.shared, k.w<float>(p.y[0]), d, d, B, 1, d, c.vocab, 0.f, 0, 1, 0, k.st));
    const size_t cache_layer = (size_t)B * Tcap * 2 * inner;
    for (int l = 0; l < Ld; ++l) {
        const auto& D = L.dec[l];
        float* y0 = k.w<float>(p.y[3 * l]);
        float* y1 = k.w<float>(p.y[3 * l + 1]);
        float* y2 = k.w<float>(p.y[3 * l + 2]);
        float* y3 = k.w<float>(p.y[3 * l + 3]);
        bf16_t* q = k.w<bf16_t>(p.dqkv_s[l]);                          // [B, inner]
        bf16_t* kc = cache + (size_t)l * cache_layer;                    // [B][Tcap][2*inner]: k | v
        bf16_t* kv = k.w<bf16_t>(p.kv_all) + (size_t)l * 2 * inner;
        RC(vlt5_layernorm_fwd(y0, k.P + D.ln_s, k.w<void>(p.yn_a[l]), nullptr, nullptr, B, d, c.eps, 0.f, 0, 0, 0, k.st));
        // q of the new token; its k,v go straight into cache slot t (row stride = one sample's cache)
        RC(k.lin_fwd(k.w<bf16_t>(p.yn_a[l]), k.Pb + D.sqkv, q, B, inner, d, 0));
        {
            vlt5_gemm_desc g;
            memset(&g, 0, sizeof g);
            g.A = k.w<bf16_t>(p.yn_a[l]); g.B = k.Pb + D.sqkv + (size_t)inner * d; g.C = kc + (size_t)t * 2 * inner;
            g.M = B; g.N = 2 * inner; g.K = d; g.lda = d; g.ldb = d; g.ldc = Tcap * 2 * inner; g.alpha = 1.f;
            RC(vlt5_gemm_bf16(&g, k.st));
        }
        RC(attn_call(k, false, q, inner, inner, kc, kc + inner, (long long)Tcap * 2 * inner, 2 * inner, k.w<bf16_t>(p.ctx_s[l]),
                     nullptr, bias_t, 1, Tk, nullptr, 0.f, 0, 1, Tk, 0));
        RC(k.lin_fwd(k.w<bf16_t>(p.ctx_s[l]), k.Pb + D.so, y1, B, d, inner, 1, 1.f, nullptr, 0, 0.f, 0, y0));
        RC(vlt5_layernorm_fwd(y1, k.P + D.ln_c, k.w<void>(p.yn_c[l]), nullptr, nullptr, B, d, c.eps, 0.f, 0, 0, 0, k.st));
        RC(k.lin_fwd(k.w<bf16_t>(p.yn_c[l]), k.Pb + D.cq, k.w<void>(p.qc[l]), B, inner, d, 0));
        RC(attn_call(k, false, k.w<bf16_t>(p.qc[l]), inner, inner, kv, kv + inner, (long long)Sx * kvw, kvw,
                     k.w<bf16_t>(p.ctx_c[l]), nullptr, nullptr, 0, 0, k.w<float>(p.mask_ext), -1e9f, 0, 1, Sx, 0));
        RC(k.lin_fwd(k.w<bf16_t>(p.ctx_c[l]), k.Pb + D.co, y2, B, d, inner, 1, 1.f, nullptr, 0, 0.f, 0, y1));
        RC(vlt5_layernorm_fwd(y2, k.P + D.ln_f, k.w<void>(p.yn_f[l]), nullptr, nullptr, B, d, c.eps, 0.f, 0, 0, 0, k.st));
        RC(k.ffn_hidden(k.w<bf16_t>(p.yn_f[l]), D.wi, k.w<bf16_t>(p.ud[l]), k.w<bf16_t>(p.hd[l]), B, 0.f, 0));
        RC(k.lin_fwd(k.w<bf16_t>(p.hd[l]), k.Pb + D.wo, y3, B, d, ff, 1, 1.f, nullptr, 0, 0.f, 0, y2));
    }
    RC(vlt5_layernorm_fwd(k.w<float>(p.y[3 * Ld]), k.P + L.dec_final_ln, k.w<void>(p.dec_out), nullptr, nullptr, B, d, c.eps, 0.f, 0,
                          0, 0, k.st));
    const float alpha = 1.0f / sqrtf((float)d);           // tied embeddings: rescale before the vocabulary projection
    RC(k.lin_fwd(k.w<bf16_t>(p.dec_out), k.Pb + L.shared, logits, B, c.vocab, d, 1, alpha));
    if (next_ids) RC(vlt5_argmax_rows(logits, B, c.vocab, next_ids, k.st));
    return VLT5_OK;
}

// The same step through the decode kernels (csrc/decode.hip): 8 launches per layer instead of 12, every one a single memory round
// trip (norms folded into the projections, k | v written straight into the cache slot, per-tile argmax in the vocabulary projection's
// epilogue).  g.done != NULL: greedy bookkeeping on the device (vlt5_decoder_step_greedy) -- then the input row of step t > 0 was left
// in the workspace by the step before, and `tokens` is read at t == 0 only.
bool decode_fast_ok(const Ctx& k) {
    const vlt5_config& c = k.c;
    if (k.tun.decode_fast == 1) return false;
    if (c.d_kv != 16 && c.d_kv != 32 && c.d_kv != 64) return false;
    if (k.p.Sx > 64 || k.s.T > 64) return false;
    return vlt5_decode_linear_supported(k.d, 1) && vlt5_decode_linear_supported(k.inner, 0) && vlt5_decode_linear_supported(k.ff, 0);
}

int decoder_step_fast(const Ctx& k, const vlt5_greedy_desc& g, bool chained) {
    const Plan& p = k.p; const Layout& L = k.lay; const vlt5_config& c = k.c; const vlt5_step& s = k.s;
    const int d = k.d, inner = k.inner, ff = k.ff, Mx = p.Mx, Sx = p.Sx, B = s.B, Tcap = s.T, Ld = c.num_decoder_layers, t = g.t;
    const int kvw = Ld * 2 * inner, Tk = t + 1;
    bf16_t* cache = (bf16_t*)g.kv_cache;
    // device-side step index (vlt5_greedy_desc.t_dev): every launch of a step t > 0 is then independent of t on the host side -- the cache
    // slot, the number of cached keys, the output column and the position of the next bias row are read from *t_dev, which the vocabulary
    // projection increments when it ends -- so the caller can capture one step in a HIP graph and replay it for every later token
    int* t_dev = g.t_dev;
    if (t == 0) {
        RC(k.wait_bucket(Ld + c.num_layers + 1));
        RC(k.wait_bucket(Ld));
        RC(vlt5_build_mask(s.input_ids, k.w<float>(p.mask_ext), B, s.L, Sx, c.pad_id, k.st));
        // cross-attention keys | values of all layers, LAYER-major for this path ([Ld][B*Sx][2*inner], one batched GEMM): a sample's
        // 58 keys of a layer are one contiguous 174 KB run instead of 128-byte pieces 36 KB apart
        vlt5_gemm_desc g;
        memset(&g, 0, sizeof g);
        g.tuning = &k.tun;
        g.A = k.w<bf16_t>(p.enc_ext); g.B = k.Pb + L.cross_kv; g.C = k.w<void>(p.kv_all); g.M = Mx; g.N = 2 * inner; g.K = d;
        g.lda = d; g.ldb = d; g.ldc = 2 * inner; g.alpha = 1.f; g.batch = Ld;
        g.batch_stride_a = 0; g.batch_stride_b = (long long)2 * inner * d; g.batch_stride_c = (long long)Mx * 2 * inner;
        RC(vlt5_gemm_bf16(&g, k.st));
        for (int l = 0; l < Ld; ++l) RC(k.wait_bucket(Ld - 1 - l));
    }
    float* bias_t = k.w<float>(p.dec_bias);                        // [H][Tcap]: relative-position bias row of query position t
    // split norms (decode.h DecLinArgs.nx_*): whoever writes a row of the residual stream also writes bf16(row * w) for the norm that reads
    // it next and the row's partial sums of squares; the projection behind the norm stages half the bytes and scales its rows by rstd.
    // One operand buffer and one partials buffer serve the whole chain (a launch reads what the launch before it wrote).
    const int nparts = d / 16;
    const bool split_norm = k.tun.decode_split_norm != 1 && (d % 64) == 0 && nparts <= 64 && vlt5_decode_linear_supported(d, 0);
    bf16_t* xn = reinterpret_cast<bf16_t*>(k.w<float>(p.tmp));
    float* xn_ssq = reinterpret_cast<float*>(reinterpret_cast<char*>(xn) + (((size_t)B * d * 2 + 255) & ~(size_t)255));
    if (t == 0 || !chained) {
        DecIoArgs io;
        memset(&io, 0, sizeof io);
        if (split_norm) { io.nx_w = k.P + L.dec[0].ln_s; io.nx_b = xn; io.nx_ssq = xn_ssq; io.nx_parts = nparts; }
        io.tokens = g.tokens; io.table = k.P + L.shared; io.d = d; io.vocab = c.vocab; io.emb_out = k.w<float>(p.y[0]);
        io.rel_table = k.P + L.dec_rel; io.lut = s.dec_lut; io.lut_ld = Tcap; io.tq = t; io.H = k.H; io.bias_out = bias_t; io.bias_ld = Tcap;
        RC(vlt5_dec_io_launch(io, B, k.st));
    }
    auto lin = [&](const float* xf, long long ln_w, const bf16_t* xb, int K, const bf16_t* W, int N, bf16_t* ob, long long ldo, float* of,
                   const float* resid, int relu) {
        DecLinArgs a;
        memset(&a, 0, sizeof a);
        a.xf = xf; a.xb = xb; a.ldx = K; a.ln_w = xf ? k.P + ln_w : nullptr; a.eps = c.eps; a.W = W; a.rows = B; a.N = N; a.K = K;
        a.alpha = 1.f; a.out_b = ob; a.ldo = ldo; a.split_col = 0x7fffffff; a.out_f = of; a.ldf = N; a.resid = resid; a.ldr = N; a.relu = relu;
        a.force_nfrag = k.tun.decode_nfrag;
        return a;
    };
    // the projection behind a norm: the bf16 operand + partials its producer left (split_norm), or the f32 row with the norm folded in
    auto normed = [&](const float* yf, long long ln_w, const bf16_t* W, int N, bf16_t* ob, long long ldo, int relu) {
        DecLinArgs a = split_norm ? lin(nullptr, 0, xn, d, W, N, ob, ldo, nullptr, nullptr, relu) : lin(yf, ln_w, nullptr, d, W, N, ob, ldo, nullptr, nullptr, relu);
        if (split_norm) { a.rs_part = xn_ssq; a.rs_n = nparts; a.eps = c.eps; }
        return a;
    };
    // the producer of residual-stream row `of` = resid + xb W^T, feeding the norm with weights next_ln
    auto resid_out = [&](const bf16_t* xb, int K, const bf16_t* W, float* of, const float* resid, long long next_ln) {
        DecLinArgs a = lin(nullptr, 0, xb, K, W, d, nullptr, 0, of, resid, 0);
        if (split_norm) { a.nx_w = k.P + next_ln; a.nx_b = xn; a.ld_nx = d; a.nx_ssq = xn_ssq; a.nx_parts = nparts; }
        return a;
    };
    const size_t cache_layer = (size_t)B * Tcap * 2 * inner;
    for (int l = 0; l < Ld; ++l) {
        const auto& D = L.dec[l];
        float* y0 = k.w<float>(p.y[3 * l]);
        float* y1 = k.w<float>(p.y[3 * l + 1]);
        float* y2 = k.w<float>(p.y[3 * l + 2]);
        float* y3 = k.w<float>(p.y[3 * l + 3]);
        bf16_t* q = k.w<bf16_t>(p.dqkv_s[l]);                          // [B, inner]
        bf16_t* kc = cache + (size_t)l * cache_layer;                    // [B][Tcap][2*inner]: k | v
        bf16_t* kv = k.w<bf16_t>(p.kv_all) + (size_t)l * Mx * 2 * inner;  // [B][Sx][2*inner]: k | v of this layer
        const int ffw = c.gated_act ? 2 * ff : ff;
        {   // norm -> q | k | v: q to its buffer, k | v into cache slot t
            DecLinArgs a = normed(y0, D.ln_s, k.Pb + D.sqkv, 3 * inner, q, inner, 0);
            a.split_col = inner; a.ldo2 = (long long)Tcap * 2 * inner;
            if (t_dev) { a.out_b2 = kc; a.t_ptr = t_dev; a.t_stride2 = 2 * inner; }
            else a.out_b2 = kc + (size_t)t * 2 * inner;
            RC(vlt5_declin_launch(a, k.st));
        }
        {
            DecCoreArgs a;
            memset(&a, 0, sizeof a);
            a.q = q; a.q_ld = inner; a.k = kc; a.v = kc + inner; a.kv_sb = (long long)Tcap * 2 * inner; a.kv_st = 2 * inner;
            a.ctx = k.w<bf16_t>(p.ctx_s[l]); a.ctx_ld = inner; a.bias = bias_t; a.bias_ld = Tcap; a.B = B; a.H = k.H; a.Tk = Tk;
            if (t_dev) { a.t_ptr = t_dev; a.Tk = 1; }
            RC(vlt5_dec_core_launch(a, c.d_kv, k.st));
        }
        {
            DecLinArgs a = resid_out(k.w<bf16_t>(p.ctx_s[l]), inner, k.Pb + D.so, y1, y0, D.ln_c);
            RC(vlt5_declin_launch(a, k.st));
        }
        {
            DecLinArgs a = normed(y1, D.ln_c, k.Pb + D.cq, inner, k.w<bf16_t>(p.qc[l]), inner, 0);
            RC(vlt5_declin_launch(a, k.st));
        }
        {
            DecCoreArgs a;
            memset(&a, 0, sizeof a);
            a.q = k.w<bf16_t>(p.qc[l]); a.q_ld = inner; a.k = kv; a.v = kv + inner; a.kv_sb = (long long)Sx * 2 * inner; a.kv_st = 2 * inner;
            a.ctx = k.w<bf16_t>(p.ctx_c[l]); a.ctx_ld = inner; a.key_mask = k.w<float>(p.mask_ext); a.mask_ld = Sx; a.mask_value = -1e9f;
            a.B = B; a.H = k.H; a.Tk = Sx;
            RC(vlt5_dec_core_launch(a, c.d_kv, k.st));
        }
        {
            DecLinArgs a = resid_out(k.w<bf16_t>(p.ctx_c[l]), inner, k.Pb + D.co, y2, y1, D.ln_f);
            RC(vlt5_declin_launch(a, k.st));
        }
        if (c.gated_act) {
            DecLinArgs a = normed(y2, D.ln_f, k.Pb + D.wi, 2 * ff, k.w<bf16_t>(p.ud[l]), 2 * ff, 0);
            RC(vlt5_declin_launch(a, k.st));
            RC(vlt5_glu_fwd(k.w<bf16_t>(p.ud[l]), k.w<bf16_t>(p.hd[l]), B, ff, 0.f, 0, k.st));
        } else {
            DecLinArgs a = normed(y2, D.ln_f, k.Pb + D.wi, ff, k.w<bf16_t>(p.hd[l]), ff, 1);
            RC(vlt5_declin_launch(a, k.st));
        }
        {
            DecLinArgs a = resid_out(k.w<bf16_t>(p.hd[l]), ff, k.Pb + D.wo, y3, y2, l + 1 < Ld ? L.dec[l + 1].ln_s : L.dec_final_ln);
            RC(vlt5_declin_launch(a, k.st));
        }
    }
    // final norm + rescale + tied lm_head (+ the first maximum of every column tile)
    const bool want_ids = g.next_ids || g.done;
    const int tiles = vlt5_declin_tiles(B, c.vocab, d, split_norm ? 2 : 1);
    float* pmax = k.w<float>(p.slab);
    int* pidx = reinterpret_cast<int*>(pmax + (size_t)B * tiles);
    if ((size_t)B * tiles * 8 > p.slab_bytes) return VLT5_ERR_ARG;
    {
        DecLinArgs a = normed(k.w<float>(p.y[3 * Ld]), L.dec_final_ln, k.Pb + L.shared, c.vocab, nullptr, 0, 0);
        a.out_f = g.logits; a.ldf = c.vocab;
        a.alpha = 1.0f / sqrtf((float)d);                           // tied embeddings: rescale before the vocabulary projection
        if (want_ids) { a.pmax = pmax; a.pidx = pidx; a.ptiles = tiles; }
        if (!g.logits && !want_ids) return VLT5_ERR_ARG;
        a.t_inc = t_dev;
        RC(vlt5_declin_launch(a, k.st));
    }
    if (want_ids) {
        DecIoArgs io;
        memset(&io, 0, sizeof io);
        io.pmax = pmax; io.pidx = pidx; io.ptiles = tiles; io.next_ids = g.next_ids;
        io.done = g.done; io.eos_id = g.eos_id; io.pad_id = g.pad_id; io.out_tokens = g.out_tokens; io.out_ld = g.out_ld; io.out_col = t + 1;
        if (t_dev) { io.t_ptr = t_dev; io.Tcap = Tcap; }
        if (chained && (t_dev || t + 1 < Tcap)) {                   // input row and bias row of the next step
            io.table = k.P + L.shared; io.d = d; io.vocab = c.vocab; io.emb_out = k.w<float>(p.y[0]);
            if (split_norm) { io.nx_w = k.P + L.dec[0].ln_s; io.nx_b = xn; io.nx_ssq = xn_ssq; io.nx_parts = nparts; }
            io.rel_table = k.P + L.dec_rel; io.lut = s.dec_lut; io.lut_ld = Tcap; io.tq = t + 1; io.H = k.H; io.bias_out = bias_t; io.bias_ld = Tcap;
        }
        RC(vlt5_dec_io_launch(io, B, k.st));
    }
    return VLT5_OK;
}

// backward (data path only) of one  x_out = x_in + drop(W_o . drop(relu(W_i . LN(x_in))))  sublayer; dx is updated in place.
// `dyd` holds bf16(dropout_out(dx)) on entry (emitted by the producer of dx); `dh` receives the hidden gradient; both are
// kept for the batched weight-gradient GEMMs at the end of the phase.  `next_dst` receives the operand of the next sublayer.
int ffn_bwd(const Ctx& k, int M, float* dx, const float* x_in, const float* rstd, const bf16_t* h, const bf16_t* u, uint32_t h_seed,
            const bf16_t* dyd, bf16_t* dh, long long wi, long long wo, long long ln, bf16_t* next_dst, uint32_t next_seed,
            bf16_t* xn_out = nullptr, const void* hbits = nullptr) {
    const Plan& p = k.p;
    const int d = k.d, ff = k.ff;
    float* tmp = k.w<float>(p.tmp);
    if (k.c.gated_act) {
        // dh_hidden = dyd Wo (scratch), then back through dropout and gelu_new(u0) * u1 into du = [du0 | du1] (kept for the wgrad)
        bf16_t* dhid = k.w<bf16_t>(p.dctx);
        RC(k.lin_dgrad(dyd, k.Pb + wo, dhid, M, d, ff, 0));
        RC(vlt5_glu_bwd(dhid, u, dh, M, ff, k.pdrop, h_seed, k.st));
    } else {
        const float gs = k.pdrop > 0.f ? drop_scale(drop_thr16(k.pdrop)) : 1.f;
        RC(k.lin_dgrad(dyd, k.Pb + wo, dh, M, d, ff, 0, 1.f, h, gs, nullptr, hbits));
    }
    int ns = 1;
    RC(k.lin_dgrad(dh, k.Pb + wi, tmp, M, k.ffw(), d, 1, 1.f, nullptr, 1.f, &ns));
    RC(k.ln_bwd(tmp, x_in, ln, rstd, dx, M, 1, 0.f, 0, 0, 0, next_dst, next_seed, ns, (long long)M * d, xn_out));
    return VLT5_OK;
}

int decoder_bwd(const Ctx& k) {
    const Plan& p = k.p; const Layout& L = k.lay; const vlt5_config& c = k.c; const vlt5_step& s = k.s;
    const int d = k.d, inner = k.inner, ff = k.ff, Md = p.Md, Mx = p.Mx, Sx = p.Sx, B = s.B, T = s.T, Ld = c.num_decoder_layers;
    const int kvw = Ld * 2 * inner;
    const float alpha = 1.0f / sqrtf((float)d);
    float* dx = k.w<float>(p.dx);
    float* tmp = k.w<float>(p.tmp);
    bf16_t* dctx = k.w<bf16_t>(p.dctx);
    bf16_t* dlog = k.w<bf16_t>(p.dlogits);
    const long long* ids = k.w<long long>(p.dec_ids);
    if (k.gnorm_ok()) HIP_RET(hipMemsetAsync(s.gnorm_partials, 0, (size_t)vlt5_gnorm_slots(&c) * sizeof(float), k.st));
    RC(vlt5_ce_bwd(k.w<float>(p.logits), s.labels, k.w<float>(p.lse_ce), s.d_loss_tok ? s.d_loss_tok : k.w<float>(p.row_w),
                   s.d_loss_tok ? nullptr : s.gout, dlog, Md, c.vocab, k.st));
    // lm_head (tied to shared): dShared = alpha * dlogits^T dec_out ; d dec_out = alpha * dlogits shared
    RC(k.lin_wgrad(dlog, c.vocab, k.w<bf16_t>(p.dec_out), d, k.Gr + L.shared, Md, c.vocab, d, alpha, 0));
    int ns_head = 1;
    RC(k.lin_dgrad(dlog, k.Pb + L.shared, tmp, Md, c.vocab, d, 1, alpha, nullptr, 1.f, &ns_head));
    RC(k.ln_bwd(tmp, k.w<float>(p.y[3 * Ld]), L.dec_final_ln, k.w<float>(p.yr[3 * Ld]), dx, Md, 0, k.pdrop, k.seed(SITE_DEC_FINAL), 0, 0,
                k.w<bf16_t>(p.d_dyd_f[Ld - 1]), k.seed(SITE_DEC_BASE + (Ld - 1) * 8 + D_FFN_OUT), ns_head, (long long)Md * d));
    for (int l = Ld - 1; l >= 0; --l) {
        const auto& D = L.dec[l];
        const uint32_t sb = SITE_DEC_BASE + l * 8;
        bf16_t* qkv = k.w<bf16_t>(p.dqkv_s[l]);
        bf16_t* kv = k.w<bf16_t>(p.kv_all) + (size_t)l * 2 * inner;
        bf16_t* dkv = k.w<bf16_t>(p.dkv_all) + (size_t)l * 2 * inner;
        bf16_t* dyd_c = k.w<bf16_t>(p.d_dyd_c[l]);
        bf16_t* dyd_s = k.w<bf16_t>(p.d_dyd_s[l]);
        bf16_t* dq_c = k.w<bf16_t>(p.d_dq_c[l]);
        bf16_t* dqkv = k.w<bf16_t>(p.d_dqkv[l]);
        RC(ffn_bwd(k, Md, dx, k.w<float>(p.y[3 * l + 2]), k.w<float>(p.yr[3 * l + 2]), k.w<bf16_t>(p.hd[l]), k.w<bf16_t>(p.ud[l]),
                   k.seed(sb + D_FFN_H), k.w<bf16_t>(p.d_dyd_f[l]), k.w<bf16_t>(p.d_dh[l]), D.wi, D.wo, D.ln_f, dyd_c, k.seed(sb + D_COUT),
                   k.fold_dec() ? k.w<bf16_t>(p.yn_f[l]) : nullptr));
        // cross-attention sublayer
        RC(k.lin_dgrad(dyd_c, k.Pb + D.co, dctx, Md, d, inner, 0));
        RC(attn_call(k, true, k.w<bf16_t>(p.qc[l]), (long long)T * inner, inner, kv, kv + inner, (long long)Sx * kvw, kvw, nullptr,
                     k.w<float>(p.lse_c[l]), nullptr, 0, 0, k.w<float>(p.mask_ext), -1e9f, 0, T, Sx, k.seed(sb + D_CPROBS), dctx, dq_c,
                     (long long)T * inner, inner, dkv, dkv + inner, (long long)Sx * kvw, kvw, nullptr));
        int ns_c = 1;
        RC(k.lin_dgrad(dq_c, k.Pb + D.cq, tmp, Md, inner, d, 1, 1.f, nullptr, 1.f, &ns_c));
        RC(k.ln_bwd(tmp, k.w<float>(p.y[3 * l + 1]), D.ln_c, k.w<float>(p.yr[3 * l + 1]), dx, Md, 1, 0.f, 0, 0, 0, dyd_s, k.seed(sb + D_SOUT),
                    ns_c, (long long)Md * d, k.fold_dec() ? k.w<bf16_t>(p.yn_c[l]) : nullptr));
        // causal self-attention sublayer
        RC(k.lin_dgrad(dyd_s, k.Pb + D.so, dctx, Md, d, inner, 0));
        RC(attn_call(k, true, qkv, (long long)T * 3 * inner, 3 * inner, qkv + inner, qkv + 2 * inner, (long long)T * 3 * inner,
                     3 * inner, nullptr, k.w<float>(p.lse_s[l]), k.w<float>(p.dec_bias), T, T, nullptr, 0.f, 1, T, T,
                     k.seed(sb + D_SPROBS), dctx, dqkv, (long long)T * 3 * inner, 3 * inner, dqkv + inner, dqkv + 2 * inner,
                     (long long)T * 3 * inner, 3 * inner, k.w<float>(p.dS_dec) + (size_t)l * B * k.H * T * T));
        int ns_s = 1;
        RC(k.lin_dgrad(dqkv, k.Pb + D.sqkv, tmp, Md, 3 * inner, d, 1, 1.f, nullptr, 1.f, &ns_s));
        RC(k.ln_bwd(tmp, k.w<float>(p.y[3 * l]), D.ln_s, k.w<float>(p.yr[3 * l]), dx, Md, 1, 0.f, 0, 0, 0,
                    l > 0 ? k.w<bf16_t>(p.d_dyd_f[l - 1]) : nullptr, l > 0 ? k.seed(SITE_DEC_BASE + (l - 1) * 8 + D_FFN_OUT) : 0u, ns_s,
                    (long long)Md * d));
    }
    RC(vlt5_relbias_bwd(k.w<float>(p.dS_dec), s.dec_lut, k.Gr + L.dec_rel, k.w<float>(p.rel_scratch), Ld * B, k.H, T, T,
                        c.rel_buckets, 0, k.st));
    RC(k.mirror_small(L.dec_rel, (long long)c.rel_buckets * k.H));
    // weight gradients of all decoder layers, one batched GEMM per weight kind, and of the stacked cross-attention K/V projection
    // -- on the side stream (if there is one) they run beside the rest of this phase and the first half of the encoder's chain
    RC(k.fork(0));
    {
        const Ctx ks = k.on_side();
        const int l1 = Ld > 1 ? 1 : 0;
        (void)l1;
        if (shadow_wgrads(k)) {
            // self-attention output projection on its own; q|k|v rides with the stacked cross-K/V gradient; the two FFN gradients ride
            // with the FFN launches of the encoder's upper half, cross o / q with those of its lower half (encoder_bwd).  (Measured
            // against it: the FFN guests cut in halves of six layers over all four FFN hosts, the 768 x 768 ones on their own -- the
            // better balance on paper gains half as much.)
            RC(dec_wgrad(ks, 4, nullptr));
            vlt5_gemm_desc guest;
            RC(dec_wgrad(ks, 5, &guest));
            RC(ks.lin_wgrad(k.w<bf16_t>(p.dkv_all), kvw, k.w<bf16_t>(p.enc_ext), d, k.Gr + L.cross_kv, Mx, kvw, d, 1.f, 0, true, &guest));
        } else {
            for (int which = 0; which < 6; ++which) RC(dec_wgrad(ks, which, nullptr));
            // cross-attention K/V projections of all layers at once
            RC(ks.lin_wgrad(k.w<bf16_t>(p.dkv_all), kvw, k.w<bf16_t>(p.enc_ext), d, k.Gr + L.cross_kv, Mx, kvw, d, 1.f, 0, true));
        }
        // gradient buckets complete here: the stacked cross-K/V projection always; the decoder layers only when all six of their
        // weight gradients ran above (shadowed: five of them are finished inside vlt5_encoder_bwd, which signals the buckets then)
        for (int b = shadow_wgrads(k) ? Ld : 0; b <= Ld; ++b) RC(ks.record(b));     // (rel-bias: before the fork)
    }
    RC(vlt5_embed_bwd(ids, dx, (long long)T * d, d, k.Gr + L.shared, B, T, d, c.vocab, k.pdrop, k.seed(SITE_DEC_EMBED), T, 0,
                       k.w<void>(p.embed_scratch), k.st));
    RC(k.lin_dgrad(k.w<bf16_t>(p.dkv_all), k.Pb + L.cross_kv, k.w<void>(p.d_enc_ext), Mx, kvw, d, 1));
    RC(k.ln_flush());
    return VLT5_OK;
}

// The encoder's weight gradients run as two batched groups: layers [cut, Le) as soon as the chain has passed layer `cut`, layers
// [0, cut) at the end.  (experiment switch: vlt5_tuning.enc_cut = number of layers in the late group)
inline int enc_cut(int Le, int tuned = 0) {
    const int c = tuned > 0 ? tuned : Le / 2;
    return c < 1 ? 1 : (c > Le - 1 ? Le - 1 : c);
}
// The six batched weight-gradient problems of the decoder stack (which: 0 FFN wo, 1 FFN wi, 2 cross o, 3 cross q, 4 self o, 5 self q|k|v):
// launched (desc == nullptr) or only described (for a grouped launch)
int dec_wgrad(const Ctx& k, int which, vlt5_gemm_desc* desc, int lo, int n, const vlt5_gemm_desc* group) {      // layers [lo, lo + n); n <= 0: all
    const Plan& p = k.p; const Layout& L = k.lay;
    const int d = k.d, inner = k.inner, ff = k.ff, Md = p.Md, Ld = k.c.num_decoder_layers;
    if (n <= 0) { lo = 0; n = Ld; }
    const int l1 = n > 1 ? lo + 1 : lo;
    switch (which) {
    case 0: return k.wgrad_batched(p.d_dyd_f[lo], p.d_dyd_f[l1], d, p.hd[lo], p.hd[l1], ff, L.dec[lo].wo, L.dec[l1].wo, n, Md, d, ff, desc, group);
    case 1: return k.wgrad_batched(p.d_dh[lo], p.d_dh[l1], k.ffw(), p.yn_f[lo], p.yn_f[l1], d, L.dec[lo].wi, L.dec[l1].wi, n, Md, k.ffw(), d, desc, group);
    case 2: return k.wgrad_batched(p.d_dyd_c[lo], p.d_dyd_c[l1], d, p.ctx_c[lo], p.ctx_c[l1], inner, L.dec[lo].co, L.dec[l1].co, n, Md, d, inner, desc, group);
    case 3: return k.wgrad_batched(p.d_dq_c[lo], p.d_dq_c[l1], inner, p.yn_c[lo], p.yn_c[l1], d, L.dec[lo].cq, L.dec[l1].cq, n, Md, inner, d, desc, group);
    case 4: return k.wgrad_batched(p.d_dyd_s[lo], p.d_dyd_s[l1], d, p.ctx_s[lo], p.ctx_s[l1], inner, L.dec[lo].so, L.dec[l1].so, n, Md, d, inner, desc, group);
    default: return k.wgrad_batched(p.d_dqkv[lo], p.d_dqkv[l1], 3 * inner, p.yn_a[lo], p.yn_a[l1], d, L.dec[lo].sqkv, L.dec[l1].sqkv, n, Md, 3 * inner, d, desc, group);
    }
}
// vlt5_step.defer_decoder_wgrads: the decoder's weight gradients (400 rows: short reductions, 45-60 us per launch on their own) ride
// in the shadow of the long 216-tile launches -- the stacked cross-K/V gradient and the encoder's FFN gradients leave 40 of the 256
// CUs idle for their whole duration; as the second problem of those grouped launches the decoder's tiles run there.
// With gradient-bucket events (data parallel) the guests are placed so that every decoder bucket is complete at the MID-POINT of the
// encoder phase (FFN wo / wi ride with the upper half's FFN launches as before, cross o + q run there as one grouped launch of their
// own instead of riding with the lower half's), and the decoder's bucket events are recorded there: one launch more than the
// single-process step instead of six, the decoder's exchange starts ~1.3 ms later but still has the lower half's chain and weight
// gradients (~2 ms at B = 80) to hide under.  vlt5_tuning.wgrad_shadow = 3: the round-4 behaviour (no shadow with events).
bool bucket_events(const Ctx& k) { return k.s.events && k.s.n_events > 0; }
// would the decoder's weight gradients ride in the encoder phase's launches?  (a function of the configuration, the tuning record and
// whether there are bucket events / a side stream only: the host asks the same question through vlt5_decoder_buckets_late)
bool shadow_rule(const vlt5_config& c, const vlt5_tuning* t, bool events, bool side) {
    const int ws = t ? t->wgrad_shadow : 0;
    if (ws == 1 || (ws == 3 && events) || side) return false;
    return c.num_layers > 1 && c.num_decoder_layers > 0;
}
bool shadow_wgrads(const Ctx& k) {
    return k.s.defer_decoder_wgrads && shadow_rule(k.c, &k.tun, bucket_events(k), k.side != nullptr);
}

// weight gradients of encoder layers [lo, hi): one batched GEMM per weight kind (grid.z = layer); guest_wo / guest_wi >= 0: that
// decoder problem (dec_wgrad) rides along with the FFN wo / wi launch
int enc_wgrads(const Ctx& k, int lo, int hi, int guest_wo = -1, int guest_wi = -1, int glo = 0, int gn = 0) {
    const Plan& p = k.p; const Layout& L = k.lay;
    const int n = hi - lo, d = k.d, inner = k.inner, ff = k.ff, M = p.M;
    if (n <= 0) return VLT5_OK;
    const int l1 = n > 1 ? lo + 1 : lo;
    vlt5_gemm_desc g_wo, g_wi;
    if (guest_wo >= 0) RC(dec_wgrad(k, guest_wo, &g_wo, glo, gn));
    if (guest_wi >= 0) RC(dec_wgrad(k, guest_wi, &g_wi, glo, gn));
    RC(k.wgrad_batched(p.e_dyd_f[lo], p.e_dyd_f[l1], d, p.h[lo], p.h[l1], ff, L.enc[lo].wo, L.enc[l1].wo, n, M, d, ff, nullptr,
                       guest_wo >= 0 ? &g_wo : nullptr));
    RC(k.wgrad_batched(p.e_dh[lo], p.e_dh[l1], k.ffw(), p.xn_f[lo], p.xn_f[l1], d, L.enc[lo].wi, L.enc[l1].wi, n, M, k.ffw(), d, nullptr,
                       guest_wi >= 0 ? &g_wi : nullptr));
    // the two attention weight gradients share one grid: 162 + 54 tiles of 256 x 256 (six layers) fill the chip only together
    const bool grouped = k.tun.wgrad_grouped != 1;
    if (grouped) {
        vlt5_gemm_desc so;
        RC(k.wgrad_batched(p.e_dyd_a[lo], p.e_dyd_a[l1], d, p.ctx[lo], p.ctx[l1], inner, L.enc[lo].so, L.enc[l1].so, n, M, d, inner, &so));
        RC(k.wgrad_batched(p.e_dqkv[lo], p.e_dqkv[l1], 3 * inner, p.xn_a[lo], p.xn_a[l1], d, L.enc[lo].sqkv, L.enc[l1].sqkv, n, M, 3 * inner, d,
                           nullptr, &so));
    } else {
        RC(k.wgrad_batched(p.e_dyd_a[lo], p.e_dyd_a[l1], d, p.ctx[lo], p.ctx[l1], inner, L.enc[lo].so, L.enc[l1].so, n, M, d, inner));
        RC(k.wgrad_batched(p.e_dqkv[lo], p.e_dqkv[l1], 3 * inner, p.xn_a[lo], p.xn_a[l1], d, L.enc[lo].sqkv, L.enc[l1].sqkv, n, M, 3 * inner, d));
    }
    return VLT5_OK;
}

int encoder_bwd(const Ctx& k) {
    const Plan& p = k.p; const Layout& L = k.lay; const vlt5_config& c = k.c; const vlt5_step& s = k.s;
    const int d = k.d, inner = k.inner, ff = k.ff, M = p.M, S = p.S, Sx = p.Sx, B = s.B, Le = c.num_layers, Ld = c.num_decoder_layers;
    float* dx = k.w<float>(p.dx);
    float* tmp = k.w<float>(p.tmp);
    bf16_t* dctx = k.w<bf16_t>(p.dctx);
    // the 2 prototype rows of every sample are detached (src/modeling_t5_our.py:615): only rows 0..S-1 flow back
    RC(k.ln_bwd(k.w<float>(p.d_enc_ext), k.w<float>(p.x[2 * Le]), L.enc_final_ln, k.w<float>(p.xr[2 * Le]), dx, M, 0, k.pdrop,
                k.seed(SITE_ENC_FINAL), S, Sx, k.w<bf16_t>(p.e_dyd_f[Le - 1]), k.seed(SITE_ENC_BASE + (Le - 1) * 8 + E_FFN_OUT)));
    for (int l = Le - 1; l >= 0; --l) {
        const auto& E = L.enc[l];
        const uint32_t sb = SITE_ENC_BASE + l * 8;
        bf16_t* qkv = k.w<bf16_t>(p.qkv[l]);
        bf16_t* dyd_a = k.w<bf16_t>(p.e_dyd_a[l]);
        bf16_t* dqkv = k.w<bf16_t>(p.e_dqkv[l]);
        RC(ffn_bwd(k, M, dx, k.w<float>(p.x[2 * l + 1]), k.w<float>(p.xr[2 * l + 1]), k.w<bf16_t>(p.h[l]), k.w<bf16_t>(p.u[l]),
                   k.seed(sb + E_FFN_H), k.w<bf16_t>(p.e_dyd_f[l]), k.w<bf16_t>(p.e_dh[l]), E.wi, E.wo, E.ln_f, dyd_a, k.seed(sb + E_ATTN_OUT),
                   k.fold_on() ? k.w<bf16_t>(p.xn_f[l]) : nullptr, k.bits_on() ? k.w<void>(p.hbits[l]) : nullptr));
        RC(k.lin_dgrad(dyd_a, k.Pb + E.so, dctx, M, d, inner, 0));
        RC(attn_call(k, true, qkv, (long long)S * 3 * inner, 3 * inner, qkv + inner, qkv + 2 * inner, (long long)S * 3 * inner,
                     3 * inner, nullptr, k.w<float>(p.lse[l]), k.w<float>(p.enc_bias), s.L, s.L, k.w<float>(p.mask), -10000.f, 0, S, S,
                     k.seed(sb + E_PROBS), dctx, dqkv, (long long)S * 3 * inner, 3 * inner, dqkv + inner, dqkv + 2 * inner,
                     (long long)S * 3 * inner, 3 * inner, k.w<float>(p.dS_enc) + (size_t)l * B * k.H * s.L * s.L));
        int ns_e = 1;
#ifdef ENC_DGRAD_HOT_A
        // experiment builds only (tools/r05_attn_bwd_fusion_bound.sh): the q|k|v input-gradient GEMM reads ONE 4.6 KB row of dq|dk|dv for every
        // row of its A operand (leading dimension 0: always cache resident) -- the consumer-side upper bound of a fusion that hands the operand
        // over on chip.  Results are WRONG.
        {
            vlt5_gemm_desc g;
            memset(&g, 0, sizeof g);
            g.tuning = &k.tun;
            g.A = dqkv; g.B = k.Pb + E.sqkv; g.C = tmp; g.M = M; g.N = d; g.K = 3 * inner; g.lda = 0; g.ldb = d; g.ldc = d; g.b_kmajor = 1;
            g.alpha = 1.f; g.out_f32 = 1;
            RC(vlt5_gemm_bf16(&g, k.st));
        }
#else
        RC(k.lin_dgrad(dqkv, k.Pb + E.sqkv, tmp, M, 3 * inner, d, 1, 1.f, nullptr, 1.f, &ns_e));
#endif
        RC(k.ln_bwd(tmp, k.w<float>(p.x[2 * l]), E.ln_s, k.w<float>(p.xr[2 * l]), dx, M, 1, 0.f, 0, 0, 0,
                    l > 0 ? k.w<bf16_t>(p.e_dyd_f[l - 1]) : nullptr, l > 0 ? k.seed(SITE_ENC_BASE + (l - 1) * 8 + E_FFN_OUT) : 0u, ns_e,
                    (long long)M * d, k.fold_enc_first(l) ? k.w<bf16_t>(p.xn_a[l]) : nullptr));
        if (Le > 1 && l == enc_cut(Le, k.tun.enc_cut)) {
            // upper half of the stack: its weight gradients are complete early, so a data-parallel all-reduce of these
            // buckets overlaps with the backward of the lower half
            RC(k.fork(1));
            const Ctx ks = k.on_side();                   // beside the lower half's chain when there is a side stream
            RC(enc_wgrads(ks, enc_cut(Le, k.tun.enc_cut), Le, shadow_wgrads(k) ? 0 : -1, shadow_wgrads(k) ? 1 : -1));   // + decoder FFN wo / wi
            if (shadow_wgrads(k) && bucket_events(k)) {
                // data parallel: the last two decoder weight gradients (cross-attention q with o as its guest) run here, so that the
                // decoder's buckets are complete -- and their exchange starts -- at the mid-point instead of at the very end
                vlt5_gemm_desc co;
                RC(dec_wgrad(ks, 2, &co));
                RC(dec_wgrad(ks, 3, nullptr, 0, 0, &co));
                for (int b = 0; b < Ld; ++b) RC(ks.record(b));
            }
            for (int b = Ld + 1; b <= Ld + 1 + (Le - 1 - l); ++b) RC(ks.record(b));
        }
    }
    RC(vlt5_relbias_bwd(k.w<float>(p.dS_enc), s.enc_lut, k.Gr + L.enc_rel, k.w<float>(p.rel_scratch), Le * B, k.H, s.L, s.L,
                        c.rel_buckets, 0, k.st));
    RC(k.mirror_small(L.enc_rel, (long long)c.rel_buckets * k.H));
    // inputs: text rows -> shared (scatter-add), visual rows -> visual embedding parameters
    RC(vlt5_embed_bwd(s.input_ids, dx, (long long)S * d, d, k.Gr + L.shared, B, s.L, d, c.vocab, k.pdrop, k.seed(SITE_ENC_EMBED), S, 0,
                       k.w<void>(p.embed_scratch), k.st));
    float* vpart = k.w<float>(p.vis_partial);
    RC(vlt5_vis_embed_bwd(dx + (size_t)s.L * d, (long long)S * d, d, k.w<float>(p.visG), k.boxes(), k.P + L.vis_wp, k.P + L.vis_bp,
                          k.P + L.vis_lnf, k.P + L.vis_lnp, k.w<float>(p.vis_rf), k.w<float>(p.vis_rp), k.w<void>(p.vis_dG), vpart,
                          k.Gr + L.shared, B, s.V, d, c.vocab, k.pdrop, k.seed(SITE_ENC_EMBED), S, s.L, k.st));
    const int nsp = vlt5_vis_embed_bwd_blocks(B * s.V);
    RC(vlt5_colsum(vpart, k.w<float>(p.small), nsp, 10 * d, 10 * d, 0, k.st));
    RC(vlt5_vis_grad_scatter(k.w<float>(p.small), k.Gr + L.vis_lnf, k.Gr + L.vis_lnp, k.Gr + L.vis_bp, k.Gr + L.vis_wp,
                             k.Gr + L.vis_img, k.Gr + L.vis_bf, d, c.n_images, k.st));
    RC(k.lin_wgrad(k.w<bf16_t>(p.vis_dG), d, k.w<bf16_t>(p.feats_bf16), c.feat_dim, k.Gr + L.vis_wf, B * s.V, d, c.feat_dim));
    RC(k.ln_flush());
    if (s.grads_bf16) {
        // bf16 staging copy of the last bucket without a pass over its 25 M elements: the two dense matrices (lm_head / shared, visual
        // feature projection) were mirrored by their weight-gradient GEMMs; what is left are the norm weights, the small visual
        // embedding parameters and the rows of `shared` the embedding backwards added to afterwards (token rows, object-order rows)
        RC(k.mirror_small(L.dec_final_ln, L.vis_wf - L.dec_final_ln));
        RC(k.mirror_small(L.vis_bf, L.shared - L.vis_bf));
        RC(vlt5_mirror_rows_bf16(k.Gr + L.shared, (bf16_t*)s.grads_bf16 + L.shared, c.vocab, d, k.w<long long>(p.dec_ids), B * s.T,
                                 s.input_ids, B * s.L, s.V < c.vocab ? s.V : c.vocab, k.st));
    }
    RC(k.record(Ld + 1 + Le));                            // embeddings + norms + visual embedding: complete BEFORE the last weight-
                                                          // gradient GEMMs, so their all-reduce hides under those (data parallel)
    // (a third flush group of Le/4 layers was measured: batches of 3 layers fill the chip too poorly, +4 % step time)
    const int low_end = Le > 1 ? enc_cut(Le, k.tun.enc_cut) : Le;
    RC(k.fork(2));
    {
        const Ctx ks = k.on_side();
        const bool guests = shadow_wgrads(k) && !bucket_events(k);
        RC(enc_wgrads(ks, 0, low_end, guests ? 2 : -1, guests ? 3 : -1));   // lower half (+ decoder cross o / q)
        for (int b = Ld + 1 + (Le - low_end); b <= Ld + Le; ++b) RC(ks.record(b));
    }
    RC(k.join(3));                                        // every gradient is complete on the caller's stream from here on
    return VLT5_OK;
}

}  // namespace

extern "C" int vlt5_encoder_late_layers(int num_layers) { return num_layers > 1 ? enc_cut(num_layers) : num_layers; }
extern "C" int vlt5_encoder_late_layers_tuned(int num_layers, const vlt5_tuning* t) {
    return num_layers > 1 ? enc_cut(num_layers, t ? t->enc_cut : 0) : num_layers;
}

// 1: with gradient-bucket events (vlt5_step.events) and vlt5_step.defer_decoder_wgrads the decoder-layer buckets [0, num_decoder_layers)
// are signalled from INSIDE vlt5_encoder_bwd (at its mid-point, together with the upper half of the encoder) -- the caller must
// enqueue its waits for them after that call; 0: they are signalled at the end of vlt5_decoder_bwd.  The stacked cross-K/V bucket
// (index num_decoder_layers) is always signalled by vlt5_decoder_bwd.
extern "C" int vlt5_decoder_buckets_late(const vlt5_config* c, const vlt5_tuning* t, int side_stream) {
    if (!c) return 0;
    return shadow_rule(*c, t, true, side_stream != 0) ? 1 : 0;
}

// The release order of the gradient buckets with events (the host cuts its waits and collectives by this) and its id.
namespace {
int release_plan(const vlt5_config& c, const vlt5_tuning* t, bool side, int* tri, int cap, int* n) {
    Layout L;
    build_layout(c, L, false);
    const int Ld = c.num_decoder_layers, Le = c.num_layers, nb = L.nbuckets;
    const int late_layers = Le > 1 ? enc_cut(Le, t ? t->enc_cut : 0) : Le;
    const int cut = Le > 1 ? Ld + 1 + (Le - late_layers) : Ld + 1;
    const bool late = shadow_rule(c, t, true, side);
    int out[15], m = 0;
    auto add = [&](int ph, int lo, int hi) { out[m * 3] = ph; out[m * 3 + 1] = lo; out[m * 3 + 2] = hi; ++m; };
    if (late) { add(0, Ld, Ld + 1); add(1, 0, Ld); } else add(0, 0, Ld + 1);
    add(1, Ld + 1, cut); add(1, nb - 1, nb); add(1, cut, nb - 1);
    if (tri) { if (cap < m) return -VLT5_ERR_ARG; for (int i = 0; i < m * 3; ++i) tri[i] = out[i]; }
    if (n) *n = m;
    unsigned h = 2166136261u;                                 // FNV-1a over the triples, folded to a positive int
    for (int i = 0; i < m * 3; ++i) { h ^= (unsigned)out[i]; h *= 16777619u; }
    return (int)(h & 0x3fffffffu) + 1;
}
}  // namespace
extern "C" int vlt5_grad_release_plan(const vlt5_config* c, const vlt5_tuning* t, int side_stream, int* triples, int cap, int* n) {
    if (!c || (triples && cap < 5)) return -VLT5_ERR_ARG;
    return release_plan(*c, t, side_stream != 0, triples, cap, n);
}

// A stream of the LOWEST priority for vlt5_step.side_stream: the weight-gradient GEMMs then only take the workgroup slots the
// input-gradient chain on the caller's (normal-priority) stream leaves free, instead of starving it.
extern "C" int vlt5_side_stream_create(void** stream) {
    if (!stream) return VLT5_ERR_ARG;
    int least = 0, greatest = 0;
    HIP_RET(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t st;
    HIP_RET(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least));
    *stream = (void*)st;
    return VLT5_OK;
}
extern "C" int vlt5_side_stream_destroy(void* stream) {
    if (!stream) return VLT5_ERR_ARG;
    HIP_RET(hipStreamDestroy((hipStream_t)stream));
    return VLT5_OK;
}

extern "C" int vlt5_layout_count(const vlt5_config* c) {
    if (!c) return -1;
    Layout L;
    build_layout(*c, L, true);
    return (int)L.e.size();
}
extern "C" int vlt5_layout_get(const vlt5_config* c, int i, char* name, int name_cap, long long* offset, int* rows, int* cols,
                               int* bucket, int* decay, int* used) {
    if (!c || !name || name_cap < 128) return VLT5_ERR_ARG;
    Layout L;
    build_layout(*c, L, true);
    if (i < 0 || i >= (int)L.e.size()) return VLT5_ERR_ARG;
    const Entry& e = L.e[i];
    snprintf(name, name_cap, "%s", e.name.c_str());
    if (offset) *offset = e.off;
    if (rows) *rows = e.rows;
    if (cols) *cols = e.cols;
    if (bucket) *bucket = e.bucket;
    if (decay) *decay = e.decay;
    if (used) *used = e.used;
    return VLT5_OK;
}
extern "C" long long vlt5_layout_total(const vlt5_config* c) {
    if (!c) return -1;
    Layout L;
    build_layout(*c, L, false);
    return L.total;
}
extern "C" long long vlt5_gnorm_slots(const vlt5_config* c) {
    if (!c) return 0;
    const int d = c->d_model, inner = c->num_heads * c->d_kv, ffw = c->gated_act ? 2 * c->d_ff : c->d_ff;
    if ((d & 63) || (inner & 63) || (ffw & 63) || (c->d_ff & 63)) return 0;
    Layout L;
    build_layout(*c, L, false);
    return (L.total + 4095) / 4096 + 1;
}
extern "C" int vlt5_layout_buckets(const vlt5_config* c) {
    if (!c) return -1;
    Layout L;
    build_layout(*c, L, false);
    return L.nbuckets;
}
extern "C" long long vlt5_workspace_bytes(const vlt5_config* c, int B, int L, int V, int T) {
    if (!c || B < 1 || L < 1 || V < 1 || T < 1) return -1;
    Plan p;
    make_plan(*c, B, L, V, T, p);
    return (long long)p.total;
}
extern "C" long long vlt5_workspace_offset(const vlt5_config* c, int B, int L, int V, int T, int which) {
    if (!c) return -1;
    Plan p;
    make_plan(*c, B, L, V, T, p);
    switch (which) {
        case VLT5_WS_ENC_OUT: return (long long)p.enc_out;
        case VLT5_WS_ENC_EXT: return (long long)p.enc_ext;
        case VLT5_WS_LOGITS: return (long long)p.logits;
        case VLT5_WS_LOSS_TOK: return (long long)p.loss_tok;
        case VLT5_WS_LOSS: return (long long)p.loss;
        case VLT5_WS_ENC_MASK_EXT: return (long long)p.mask_ext;
        case VLT5_WS_DEC_OUT: return (long long)p.dec_out;
        default: return -1;
    }
}

int vlt5_build_flags_engine() {
#ifdef ENC_DGRAD_HOT_A
    return VLT5_BUILD_ENC_DGRAD_HOT_A;
#else
    return 0;
#endif
}
// the caller's release plan against what this backward call is about to do (vlt5_step.release_plan_id)
static int check_release_plan(const Ctx& k) {
    if (!k.s.release_plan_id || !bucket_events(k)) return VLT5_OK;
    if (!k.s.defer_decoder_wgrads) return VLT5_ERR_PLAN;          // (the plan describes the deferred order)
    return release_plan(k.c, &k.tun, k.side != nullptr, nullptr, 0, nullptr) == k.s.release_plan_id ? VLT5_OK : VLT5_ERR_PLAN;
}
#define ENGINE_ENTRY(fn, bwd)                                                   \
    extern "C" int vlt5_##fn(const vlt5_config* c, const vlt5_step* s, void* stream) { \
        if (!c || !s) return VLT5_ERR_ARG;                                      \
        Ctx k(*c, *s, stream);                                                  \
        int rc = k.check(bwd);                                                  \
        if (rc) return rc;                                                      \
        if (bwd) { rc = check_release_plan(k); if (rc) return rc; }             \
        return fn(k);                                                           \
    }
extern "C" int vlt5_decoder_step(const vlt5_config* c, const vlt5_step* s, const long long* tokens, int t, void* kv_cache,
                                 float* logits, long long* next_ids, void* stream) {
    if (!c || !s || !tokens || !kv_cache || !logits) return VLT5_ERR_ARG;
    if (t < 0 || t >= s->T || s->training) return VLT5_ERR_ARG;
    Ctx k(*c, *s, stream);
    int rc = k.check(false);
    if (rc) return rc;
    if (!s->dec_lut || !s->input_ids) return VLT5_ERR_ARG;
    if (decode_fast_ok(k)) {
        vlt5_greedy_desc g;
        memset(&g, 0, sizeof g);
        g.tokens = tokens; g.t = t; g.kv_cache = kv_cache; g.logits = logits; g.next_ids = next_ids;
        return decoder_step_fast(k, g, false);
    }
    return decoder_step(k, tokens, t, (bf16_t*)kv_cache, logits, next_ids);
}
extern "C" int vlt5_decoder_step_greedy(const vlt5_config* c, const vlt5_step* s, const vlt5_greedy_desc* g, void* stream) {
    if (!c || !s || !g || !g->kv_cache || !g->done || !g->out_tokens) return VLT5_ERR_ARG;
    if (g->t_dev) { if (s->training || g->out_ld < s->T) return VLT5_ERR_ARG; }              // (the step index lives on the device: t only says "first step or not")
    else if (g->t < 0 || g->t >= s->T || s->training || g->out_ld < g->t + 2) return VLT5_ERR_ARG;
    if (g->t == 0 && !g->tokens) return VLT5_ERR_ARG;
    Ctx k(*c, *s, stream);
    int rc = k.check(false);
    if (rc) return rc;
    if (!s->dec_lut || !s->input_ids) return VLT5_ERR_ARG;
    if (!decode_fast_ok(k)) return VLT5_ERR_ARG;                  // (the caller falls back to vlt5_decoder_step + its own bookkeeping)
    return decoder_step_fast(k, *g, true);
}
extern "C" int vlt5_decode_fast_supported(const vlt5_config* c, const vlt5_step* s) {
    if (!c || !s) return 0;
    Ctx k(*c, *s, nullptr);
    return k.check(false) == VLT5_OK && decode_fast_ok(k) ? 1 : 0;
}
ENGINE_ENTRY(encoder_fwd, false)
ENGINE_ENTRY(decoder_fwd, false)
ENGINE_ENTRY(decoder_bwd, true)
ENGINE_ENTRY(encoder_bwd, true)
