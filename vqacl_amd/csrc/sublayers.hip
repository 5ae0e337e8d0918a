// Sublayer entry points of SURVEY 8(b) that are compositions of the kernels in this library: the feed-forward sublayer, the backward
// of the two decoder attention sublayers, and the tied vocabulary projection with its cross-entropy.  Pure host code: each entry only
// enqueues launches on the caller's stream; every buffer incl. the scratch belongs to the caller.
//
// Reference being replaced (paths relative to the reference's VL-T5/ directory, "HF" = transformers 4.2.1 models/t5/modeling_t5.py):
//   vlt5_ffn_fwd / _bwd              HF T5LayerFF.forward (T5LayerNorm -> T5DenseReluDense / T5DenseGatedGeluDense -> dropout -> residual)
//                                    as called from src/modeling_t5_our.py:282-293 (encoder T5Block) and :641-655 (decoder), and
//                                    autograd's backward of it (src/vqacl.py:461)
//   vlt5_dec_self_attn_bwd / vlt5_cross_attn_bwd   backward of HF T5Attention inside T5LayerSelfAttention / T5LayerCrossAttention of the
//                                    decoder (the part between the norm and the residual, the counterpart of vlt5_dec_self_attn_fwd /
//                                    vlt5_cross_attn_fwd)
//   vlt5_lmhead_ce_fwd / _bwd        src/modeling_t5_our.py:661-686: sequence_output * d_model^-0.5 -> lm_head (tied to shared) ->
//                                    CrossEntropyLoss(ignore_index=-100, reduction='none'), and its backward
#include <math.h>
#include <string.h>
#include "common.h"
#include "vlt5_hip.h"

namespace {
inline long long up256(long long x) { return (x + 255) / 256 * 256; }

struct Gemm {
    void* stream;
    const vlt5_tuning* tuning;
    vlt5_gemm_desc m;
    // C[M,N] = epi(alpha * A B^T) with the operand orders of the three passes (forward: both row-major; dgrad: B k-major; wgrad: both k-major)
    vlt5_gemm_desc& base(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int akm, int bkm, int f32) {
        memset(&m, 0, sizeof m);
        m.A = A; m.B = B; m.C = C; m.M = M; m.N = N; m.K = K; m.lda = lda; m.ldb = ldb; m.ldc = ldc;
        m.a_kmajor = akm; m.b_kmajor = bkm; m.alpha = 1.f; m.out_f32 = f32; m.tuning = tuning;
        return m;
    }
    int run() { return vlt5_gemm_bf16(&m, stream); }
};
}  // namespace

// ---- feed-forward sublayer ------------------------------------------------------------------------------------------------------
extern "C" long long vlt5_ffn_bwd_workspace_bytes(int M, int d_model, int d_ff, int gated) {
    if (M < 1 || d_model < 1 || d_ff < 1) return -1;
    const long long ffw = gated ? 2ll * d_ff : d_ff;
    return up256((long long)M * d_model * 2) + up256((long long)M * ffw * 2) + up256((long long)M * d_ff * 2) + up256((long long)M * d_model * 4) +
           up256((long long)vlt5_layernorm_bwd_blocks(M) * d_model * 4);
}

extern "C" int vlt5_ffn_fwd(const vlt5_ffn_desc* f, void* stream) {
    if (!f || !f->x || !f->ln_w || !f->wi_bf16 || !f->wo_bf16 || !f->x_out || !f->xn_bf16 || !f->rstd || !f->h_bf16) return VLT5_ERR_ARG;
    if (f->M < 1 || f->d_model < 1 || f->d_ff < 1 || (f->gated && !f->u_bf16)) return VLT5_ERR_ARG;
    if ((f->d_model & 7) || (f->d_ff & 7)) return VLT5_ERR_ALIGN;
    const int M = f->M, d = f->d_model, ff = f->d_ff;
    int rc = vlt5_layernorm_fwd(f->x, f->ln_w, f->xn_bf16, nullptr, f->rstd, M, d, f->eps, 0.f, 0, 0, 0, stream);
    if (rc) return rc;
    Gemm g{stream, f->tuning};
    if (f->gated) {                        // u = xn [wi_0; wi_1]^T, h = dropout(gelu_new(u0) * u1)
        g.base(f->xn_bf16, f->wi_bf16, f->u_bf16, M, 2 * ff, d, d, d, 2 * ff, 0, 0, 0);
        if ((rc = g.run())) return rc;
        if ((rc = vlt5_glu_fwd(f->u_bf16, f->h_bf16, M, ff, f->drop_p, f->seed_hidden, stream))) return rc;
    } else {                               // h = dropout(relu(xn Wi^T)) in the GEMM's epilogue
        vlt5_gemm_desc& m = g.base(f->xn_bf16, f->wi_bf16, f->h_bf16, M, ff, d, d, d, ff, 0, 0, 0);
        m.relu = 1; m.drop_p = f->drop_p; m.drop_seed = f->seed_hidden;
        if ((rc = g.run())) return rc;
    }
    vlt5_gemm_desc& m = g.base(f->h_bf16, f->wo_bf16, f->x_out, M, d, ff, ff, ff, d, 0, 0, 1);       // x_out = x + dropout(h Wo^T)
    m.resid = f->x; m.ldr = d; m.drop_p = f->drop_p; m.drop_seed = f->seed_out;
    return g.run();
}

extern "C" int vlt5_ffn_bwd(const vlt5_ffn_desc* f, const vlt5_ffn_grads* q, void* workspace, void* stream) {
    if (!f || !q || !workspace || !f->x || !f->ln_w || !f->wi_bf16 || !f->wo_bf16 || !f->xn_bf16 || !f->rstd || !f->h_bf16 || !q->dy || !q->dx ||
        !q->d_wi || !q->d_wo || !q->d_ln_w || (f->gated && !f->u_bf16))
        return VLT5_ERR_ARG;
    const int M = f->M, d = f->d_model, ff = f->d_ff, ffw = f->gated ? 2 * ff : ff;
    char* ws = (char*)workspace;
    bf16_t* dyd = (bf16_t*)ws;   ws += up256((long long)M * d * 2);
    bf16_t* dh = (bf16_t*)ws;    ws += up256((long long)M * ffw * 2);
    bf16_t* dhid = (bf16_t*)ws;  ws += up256((long long)M * ff * 2);
    float* dxn = (float*)ws;     ws += up256((long long)M * d * 4);
    float* lnpart = (float*)ws;
    int rc = vlt5_drop_cast(q->dy, dyd, M, d, f->drop_p, f->seed_out, stream);          // dyd = bf16(dropout'(dy))
    if (rc) return rc;
    Gemm g{stream, f->tuning};
    if (f->gated) {
        g.base(dyd, f->wo_bf16, dhid, M, ff, d, d, ff, ff, 0, 1, 0);
        if ((rc = g.run())) return rc;
        if ((rc = vlt5_glu_bwd(dhid, f->u_bf16, dh, M, ff, f->drop_p, f->seed_hidden, stream))) return rc;
    } else {                               // dh = (dyd Wo) gated by the saved post-ReLU, post-dropout activation
        vlt5_gemm_desc& m = g.base(dyd, f->wo_bf16, dh, M, ff, d, d, ff, ff, 0, 1, 0);
        m.gate = f->h_bf16; m.ldg = ff; m.gate_scale = f->drop_p > 0.f ? drop_scale(drop_thr16(f->drop_p)) : 1.f;
        if ((rc = g.run())) return rc;
    }
    g.base(dyd, f->h_bf16, q->d_wo, d, ff, M, d, ff, ff, 1, 1, 1);                        // dWo [d, ff] = dyd^T h
    if ((rc = g.run())) return rc;
    g.base(dh, f->xn_bf16, q->d_wi, ffw, d, M, ffw, d, d, 1, 1, 1);                       // dWi [ffw, d] = dh^T xn
    if ((rc = g.run())) return rc;
    g.base(dh, f->wi_bf16, dxn, M, d, ffw, ffw, d, d, 0, 1, 1);                           // dxn = dh Wi
    if ((rc = g.run())) return rc;
    if (q->dx != q->dy) HIP_RET(hipMemcpyAsync(q->dx, q->dy, (size_t)M * d * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return vlt5_layernorm_bwd(dxn, f->x, f->ln_w, f->rstd, q->dx, q->d_ln_w, lnpart, M, d, 1, 0, 0.f, 0, 0, 0, nullptr, 0.f, 0, stream);
}

// ---- backward of the decoder attention sublayers (between the norm and the residual) --------------------------------------------------
extern "C" long long vlt5_dec_attn_bwd_workspace_bytes(int B, int Tq, int H, int d_kv, int d_model) {
    if (B < 1 || Tq < 1 || H < 1 || d_kv < 1 || d_model < 1) return -1;
    const long long M = (long long)B * Tq, inner = (long long)H * d_kv;
    return up256(M * d_model * 2) + up256(M * inner * 2);
}

static int dec_attn_bwd(const vlt5_dec_attn_desc* e, const vlt5_dec_attn_grads* q, void* workspace, void* stream, bool cross) {
    if (!e || !q || !workspace || !e->xn_bf16 || !e->w_bf16 || !e->wo_bf16 || !e->proj_bf16 || !q->d_out || !q->d_xn || !q->d_w || !q->d_wo ||
        !q->d_proj || !e->core.ctx || !e->core.lse)
        return VLT5_ERR_ARG;
    if (cross && (!q->dk || !q->dv)) return VLT5_ERR_ARG;
    const vlt5_attn_desc& c = e->core;
    const int B = c.B, Tq = c.Tq, H = c.H, dk = c.dk, dm = e->d_model, M = B * Tq, inner = H * dk, pw = cross ? inner : 3 * inner;
    char* ws = (char*)workspace;
    bf16_t* dob = (bf16_t*)ws;   ws += up256((long long)M * dm * 2);
    bf16_t* dctx = (bf16_t*)ws;
    int rc = vlt5_cast_bf16(q->d_out, dob, (long long)M * dm, stream);                     // (the sublayer's dropout sits behind the slab sum: the caller's)
    if (rc) return rc;
    Gemm g{stream, nullptr};
    g.base(dob, e->wo_bf16, dctx, M, inner, dm, dm, inner, inner, 0, 1, 0);              // d ctx = d_out Wo
    if ((rc = g.run())) return rc;
    g.base(dob, c.ctx, q->d_wo, dm, inner, M, dm, inner, inner, 1, 1, 1);                // dWo [d_model, inner] = d_out^T ctx
    if ((rc = g.run())) return rc;
    vlt5_attn_desc a = c;
    bf16_t* dp = (bf16_t*)q->d_proj;
    a.d_ctx = dctx; a.do_sb = (long long)Tq * inner; a.do_st = inner;
    if (cross) {
        a.dq = dp; a.dq_sb = (long long)Tq * inner; a.dq_st = inner;
        a.dk_ = q->dk; a.dv = q->dv; a.dk_sb = a.dv_sb = q->dkv_sb; a.dk_st = a.dv_st = q->dkv_st;
    } else {
        a.dq = dp; a.dk_ = dp + inner; a.dv = dp + 2 * inner;
        a.dq_sb = a.dk_sb = a.dv_sb = (long long)Tq * 3 * inner; a.dq_st = a.dk_st = a.dv_st = 3 * inner;
    }
    a.dbias = q->d_scores;
    if ((rc = vlt5_attn_bwd(&a, stream))) return rc;
    g.base(dp, e->xn_bf16, q->d_w, pw, dm, M, pw, dm, dm, 1, 1, 1);                       // dW [pw, d_model] = dproj^T xn
    if ((rc = g.run())) return rc;
    g.base(dp, e->w_bf16, q->d_xn, M, dm, pw, pw, dm, dm, 0, 1, 1);                       // d xn = dproj W
    return g.run();
}
extern "C" int vlt5_dec_self_attn_bwd(const vlt5_dec_attn_desc* e, const vlt5_dec_attn_grads* q, void* workspace, void* stream) {
    return dec_attn_bwd(e, q, workspace, stream, false);
}
extern "C" int vlt5_cross_attn_bwd(const vlt5_dec_attn_desc* e, const vlt5_dec_attn_grads* q, void* workspace, void* stream) {
    return dec_attn_bwd(e, q, workspace, stream, true);
}

// ---- rescale + tied lm_head + cross-entropy ---------------------------------------------------------------------------------------------
extern "C" int vlt5_lmhead_ce_fwd(const vlt5_lmhead_ce_desc* h, void* stream) {
    if (!h || !h->x_bf16 || !h->emb_bf16 || !h->labels || !h->logits || !h->loss_tok || !h->lse || h->rows < 1 || h->d_model < 1 || h->vocab < 1)
        return VLT5_ERR_ARG;
    if ((h->d_model & 7) || (h->vocab & 7)) return VLT5_ERR_ALIGN;
    Gemm g{stream, h->tuning};
    vlt5_gemm_desc& m = g.base(h->x_bf16, h->emb_bf16, h->logits, h->rows, h->vocab, h->d_model, h->d_model, h->d_model, h->vocab, 0, 0, 1);
    m.alpha = 1.0f / sqrtf((float)h->d_model);           // tied embeddings: rescale before the projection (modeling_t5_our.py:661-668)
    int rc = g.run();
    if (rc) return rc;
    return vlt5_ce_fwd(h->logits, h->labels, h->loss_tok, h->lse, h->rows, h->vocab, stream);
}
extern "C" int vlt5_lmhead_ce_bwd(const vlt5_lmhead_ce_desc* h, const vlt5_lmhead_ce_grads* q, void* stream) {
    if (!h || !q || !h->x_bf16 || !h->emb_bf16 || !h->labels || !h->logits || !h->lse || !q->d_loss_tok || !q->dlogits_bf16 || !q->d_x ||
        !q->d_emb)
        return VLT5_ERR_ARG;
    const int R = h->rows, V = h->vocab, d = h->d_model;
    int rc = vlt5_ce_bwd(h->logits, h->labels, h->lse, q->d_loss_tok, nullptr, q->dlogits_bf16, R, V, stream);
    if (rc) return rc;
    const float alpha = 1.0f / sqrtf((float)d);
    Gemm g{stream, h->tuning};
    vlt5_gemm_desc& a = g.base(q->dlogits_bf16, h->x_bf16, q->d_emb, V, d, R, V, d, d, 1, 1, 1);          // dE [V, d] = alpha dlogits^T x
    a.alpha = alpha; a.accum = q->accum_d_emb;
    if ((rc = g.run())) return rc;
    vlt5_gemm_desc& b = g.base(q->dlogits_bf16, h->emb_bf16, q->d_x, R, d, V, V, d, d, 0, 1, 1);           // d x = alpha dlogits E
    b.alpha = alpha;
    return g.run();
}
