/* vlt5_hip.h -- C ABI of libvlt5_hip.so, the MI355X (gfx950) kernels for the VQACL VL-T5 hot path.
 *
 * The reference (zhangxi1997/VQACL) is pure Python and has no FFI of its own: every entry point
 * here replaces the eager PyTorch op sequence named next to it (paths relative to the reference's
 * VL-T5/ directory; "HF" = transformers 4.2.1 models/t5/modeling_t5.py, the pinned dependency).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (the engine never allocates); `stream` is a hipStream_t passed as void*.
 *   - all functions are stream-ordered and re-entrant; one stream per calling thread.  The only process state: per-kernel,
 *     per-device "large LDS enabled" flags (atomic) and one measurement hook that the product path never turns on -- the GEMM
 *     timing recorder (vlt5_gemm_timing_*, not thread-safe).  The library reads NO environment variables: experiment switches
 *     travel in a vlt5_tuning record the caller passes (vlt5_step.tuning, vlt5_gemm_desc.tuning; NULL = the defaults).
 *   - return 0 on success, VLT5_ERR_* for argument errors, otherwise the hipError_t of the launch.
 *     Nothing throws across the ABI.
 *   - "bf16" tensors are raw 16-bit bfloat16; contiguous (feature) dimensions must be multiples of 8.
 */
#ifndef VLT5_HIP_H
#define VLT5_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VLT5_OK 0
#define VLT5_ERR_ARG 1001
#define VLT5_ERR_ALIGN 1002
#define VLT5_ERR_PLAN 1003   /* vlt5_step.release_plan_id does not name the order THIS call completes gradient buckets in */
#define VLT5_ABI_VERSION 8

int vlt5_abi_version(void);
/* 0 for the product build.  Non-zero: the library was compiled with experiment switches whose RESULTS ARE WRONG on purpose (upper-bound
 * measurements: tools/r05_attn_bwd_fusion_bound.sh) or that stamp timelines into user buffers -- a loader must refuse such a build
 * unless it was asked for one (vqacl_amd/_lib.py: VLT5_ALLOW_EXPERIMENT=1). */
#define VLT5_BUILD_ENC_DGRAD_HOT_A   1
#define VLT5_BUILD_ATTN_BWD_NO_STORE 2
#define VLT5_BUILD_TIMELINE          4
int vlt5_build_flags(void);

/* ---- experiment switches (A/B runs, tests of the alternative paths) -------------------------------------------------------
 * Every field: 0 = the library's default; the product path passes NULL or an all-zero record.  Read by the call that receives it,
 * never stored -- no process state.  (The host side fills it once from VLT5_* environment variables: vqacl_amd/_lib.py.) */
typedef struct {
    int fold_norm;          /* 1: encoder RMS norms as stand-alone launches (unfolded); 2: folded (default) */
    int fold_norm_dec;      /* 2: also fold the decoder's cross / FFN norms (default 1: off) */
    int fused_attn;         /* 1: encoder q|k|v GEMM + attention core as separate launches; 2: the fused kernel; 0 (default): the fused
                               kernel where its grid fills the chip (B / 2 x H / 2 >= 128 workgroups), the separate launches below that */
    int fused_heads;        /* heads per workgroup of the fused encoder kernel: 1 or 2 (default 2) */
    int dec_fused;          /* 2: fused decoder attention sublayers (csrc/dec_attn.hip) in the training forward (default 1: off) */
    int enc_cut;            /* > 0: number of encoder layers whose weight gradients run in the late group (default num_layers / 2) */
    int wgrad_shadow;       /* 1: decoder weight gradients as launches of their own; 2: in the shadow of the encoder's (default; with
                               gradient-bucket events the decoder's buckets are then signalled from the encoder phase's mid-point launch
                               that finishes them); 3: in the shadow only without bucket events, launches of their own with them (the
                               decoder's buckets are released at the end of the decoder phase: the round-4 behaviour) */
    int wgrad_grouped;      /* 1: the two attention weight gradients of a layer group as two launches; 2: one grouped launch (default) */
    int gemm_t128_kmkm;     /* > 0: tile-count threshold of the 128x128 tile for k-major/k-major problems (default 100) */
    int gemm_t256_km;       /* > 0: ... of the 256x256 tile for a k-major A operand (default 160) */
    int gemm_t256_min;      /* > 0: ... of the 8-wave tiles for a row-major A operand (default 100) */
    int gemm_dec_tall;      /* 1: 64x64 tiles for the wide small-M input gradients; 2: 128x64 (default) */
    int gemm_split_kmin;    /* > 0: shortest reduction the automatic split-K cuts (default 768) */
    int decode_fast;        /* 1: greedy decoding steps through the tiled GEMM / attention launches of the training path;
                               2: through the decode kernels (csrc/decode.hip; default where the shapes allow) */
    int decode_split_norm;  /* decode kernels, the T5 norm in front of a projection: 1: folded into the projection (f32 rows in, norm weight and sum
                               of squares inside); 2: split -- the launch that writes a residual-stream row also writes bf16(row * w) and the row's
                               partial sums of squares, the projection reads half the bytes (default) */
    int gemm_rmkm_tile;     /* row-major A, k-major B (input gradients), the >= 256-tile 4-wave branch: 1: 128x64 (tall, round-1 choice);
                               2: 64x128 (round-5 step-faithful sweep: 34.1 -> 30.4 us warm on 4480x768x3072); 0: the default */
    int gemm_rmrm_f32_tile; /* row-major A and B with an f32 output (sublayer outputs), same branch: 1: 64x128; 2: 128x64 -- only where the
                               consumer of a folded norm's partials takes 24 of them (not in front of the fused encoder attention kernel) */
    int gemm_split_cap;     /* > 0: largest automatic split-K factor of the small-output policy (default 4; 8 until round 5) */
    int decode_nfrag;       /* decode kernels: 1, 2 or 4 forces the column-tile width of the projections to 16 x this many columns; 16: the
                               round-4 geometry rule (widest tile that fills the chip) at every row count; 0: the launch geometry's own choice */
    int ffn_gate_bits;      /* 1: the encoder's FFN hidden gradient gates by the saved bf16 activation (27.5 MB per layer at B = 80, read cold);
                               2 / 0 (default): by the ReLU sign bits the forward projection leaves (vlt5_gemm_desc.relu_bits_out / gate_bits) */
} vlt5_tuning;

/* ---- GEMM: C[M,N] = epi(alpha * sum_k A[m,k] B[n,k]) -------------------------------------------
 * replaces nn.Linear in T5Attention q/k/v/o (HF T5Attention.forward), T5DenseReluDense wi/wo,
 * VisualEmbedding.feat_embedding[0] (src/modeling_t5_our.py:107), lm_head (:671) and, with the
 * k-major flags, their autograd dgrad / wgrad matmuls (src/vqacl.py:461 loss.backward()). */
typedef struct vlt5_gemm_desc_s {
    const void* A; const void* B; void* C;   /* A,B bf16; C bf16 or f32 (out_f32) */
    int M, N, K;
    int lda, ldb, ldc;                       /* leading dimensions in elements */
    int a_kmajor, b_kmajor;                  /* 0: element (r,k) at r*ld+k ; 1: at k*ld+r */
    float alpha;
    const float* bias;                       /* [N] or NULL */
    const float* resid; int ldr;             /* f32 [M,ldr] added after dropout, or NULL */
    const void* gate; int ldg; float gate_scale; /* bf16 [M,ldg]: v = gate>0 ? v*gate_scale : 0 */
    float drop_p; uint32_t drop_seed;        /* inverted dropout on element index m*N+n */
    int relu, out_f32, accum;                /* accum: C += (f32 only) */
    int split_k; void* workspace;            /* split_k>1: f32 slabs [split_k][M*ldc], plain epilogue, ldc==N */
    int tile_m, tile_n;                      /* 0 = heuristic; else 64x64, 64x128, 128x64, 128x128, 256x256 or (row-major A) 224x256, 160x256 */
    int batch;                               /* > 1: batch of equal-shaped GEMMs (grid.z); entry z uses A + z*batch_stride_a, ... */
    long long batch_stride_a, batch_stride_b, batch_stride_c;   /* element strides (may be negative); resid uses batch_stride_c */
    int defer_reduce;                        /* split_k>1: leave the f32 slabs in `workspace` (slab s = partial sum of k-slice s) for
                                                a consumer that sums them itself (vlt5_layernorm_bwd_slabs); C is not written */
    int split_used;                          /* out: the number of slabs actually written (<= split_k), 1 if not split */
    void* c_bf16_copy;                       /* optional: plain f32 output (no epilogue option, no accum) is ALSO written rounded to bf16
                                                here, same ldc / batch stride -- the staging copy of a data-parallel gradient bucket */
    const struct vlt5_gemm_desc_s* grouped_with; /* optional: a SECOND problem launched in the same (flat) grid -- plain f32 output, same
                                                operand orders and alpha, no split; its own shape, reduction length and batch count:
                                                weight gradients whose tile counts fill the chip only together, or a short problem
                                                in the shadow of a long one.  Its own grouped_with / tile fields are ignored */
    /* T5 RMS norm (HF T5LayerNorm.forward: x * rsqrt(mean(x^2) + eps) * w, no mean, no bias) folded AROUND the GEMMs instead of
     * launched between them: LN(x) W^T = rstd (.) ((x (.) w) W^T).
     * producer (f32 output with residual, no bias / relu / gate / split-K; N <= 1024): the finished rows x are ALSO written as
     *   bf16(x * emit_norm_w[n]) to emit_xw_bf16 (same ldc) and every 32- / 64-column slice of a row leaves its sum of squares in
     *   emit_partials[m * 32 + slice]; emit_nparts (out) = slices per row.
     * consumer: A = that bf16 operand; every output row m is scaled by rstd[m] = rsqrt(sum_{s < norm_nparts} norm_partials[m*32+s]
     *   / norm_d + norm_eps) before the rest of the epilogue, and rstd[m] is stored to norm_rstd_out (for the norm's backward). */
    const float* emit_norm_w; void* emit_xw_bf16; float* emit_partials; int emit_nparts;
    const float* norm_partials; int norm_nparts, norm_d; float norm_eps; float* norm_rstd_out;
    float* sumsq; long long sumsq_batch_stride; /* optional (plain f32 output, no split): sum of squares of every output tile, written to
                                                sumsq[z * sumsq_batch_stride + t], t < tiles of the launch's tile shape (at most
                                                ceil(M/64)*ceil(N/64)); fixed reduction order -- the optimizer's gradient norm
                                                without a second pass over the gradients (vlt5_gnorm_finish) */
    const vlt5_tuning* tuning;               /* optional experiment switches of the tile policy (NULL = defaults) */
    /* ReLU sign bits instead of the saved activation (round 6): the backward of y = dropout(relu(x W^T)) gates dL/dh by h > 0 -- one BIT
     * per element.  relu_bits_out (with relu, bf16 output, N % 8 == 0, no batch / split): the epilogue ALSO writes bit (n & 7) of byte
     * relu_bits_out[m * ld_bits + n / 8] = (the stored bf16 value != 0).  gate_bits (instead of `gate`, same rules): v = bit ? v *
     * gate_scale : 0 -- the hidden-gradient GEMM of an encoder FFN reads 1.7 MB of bits instead of 27.5 MB of cold activations. */
    void* relu_bits_out; const void* gate_bits; int ld_bits;   /* ld_bits: bytes per row of the bit matrix, >= N / 8 */
} vlt5_gemm_desc;
int vlt5_gemm_bf16(vlt5_gemm_desc* d, void* stream);     /* writes d->split_used */
long long vlt5_gemm_workspace_bytes(int M, int ldc, int split_k);
/* the split-K factor the engine uses for a plain f32 output [M,N] reduced over Kred (1 = no split) */
int vlt5_gemm_auto_split(int M, int N, int Kred, long long slab_bytes);
int vlt5_gemm_auto_split_tuned(int M, int N, int Kred, long long slab_bytes, const vlt5_tuning* tuning);

/* measurement hook for bench.py's roofline (no counterpart in the reference): while enabled (max_launches > 0; 0 disables and
 * frees), every GEMM kernel dispatch carries its own start/stop HIP events; collect() waits for them and returns one record per
 * dispatch in launch order (the number of records, -1 on error), then resets.  Process-global, not thread-safe. */
typedef struct { int M, N, K, batch, tile_m, tile_n, a_kmajor, b_kmajor, splits, workgroups, out_f32; float ms;
                 int M2, N2, K2, batch2; /* the second problem of a grouped launch, 0 otherwise */ } vlt5_gemm_timing_rec;
int vlt5_gemm_timing_enable(int max_launches);
int vlt5_gemm_timing_collect(vlt5_gemm_timing_rec* out, int cap);

/* ---- T5LayerNorm (RMS, no mean, no bias): HF T5LayerNorm.forward ------------------------------
 * y = x * rsqrt(mean(x^2)+eps) * w, statistics in f32.  Optional inverted dropout on y
 * (encoder/decoder final norm, src/modeling_t5_our.py:314-315).  Output row r is written at row
 * (r / out_group) * out_group_stride + r % out_group  (out_group = 0: same row). */
int vlt5_layernorm_fwd(const float* x, const float* w, void* y_bf16, float* y_f32, float* rstd,
                       int rows, int d, float eps, float drop_p, uint32_t drop_seed,
                       int out_group, int out_group_stride, void* stream);
/* dx (+)= d/dx; dy row r is read at the remapped row as above (in_group*).
 * dw_partial: f32 [vlt5_layernorm_bwd_blocks(rows)][d] per-workgroup partial sums of the weight gradient.
 * dw: if non-NULL a second launch reduces the partials into dw (+= if accum_dw) in a fixed order; if NULL the caller
 *   reduces later (the engine reduces all norms of a phase with ONE vlt5_colsum_multi launch).
 * dx_bf16 (optional): also emit bf16(dropout(dx)) with (dx_drop_p, dx_drop_seed), element index r*d+c -- the operand the
 *   next sublayer's backward GEMMs read, saving a separate vlt5_drop_cast pass. */
/* same, with the input row assembled first from the `nslabs` split-K slabs of the producing GEMM (vlt5_gemm_desc.defer_reduce):
 * x_out[r] = resid[r] + dropout(sum_s slabs[s*slab_stride + r*d ..], resid_drop_p, resid_drop_seed) (element index r*d+c, the
 * index the GEMM's own dropout epilogue uses), then normalised as above -- the kernel stands in for the residual/dropout
 * epilogue and the slab reduction of that GEMM */
int vlt5_layernorm_fwd_slabs(const float* slabs, int nslabs, long long slab_stride, const float* resid, float* x_out,
                             float resid_drop_p, uint32_t resid_drop_seed, const float* w, void* y_bf16, float* y_f32,
                             float* rstd, int rows, int d, float eps, float drop_p, uint32_t drop_seed, int out_group,
                             int out_group_stride, void* stream);
int vlt5_layernorm_bwd(const float* dy, const float* x, const float* w, const float* rstd,
                       float* dx, float* dw, float* dw_partial, int rows, int d,
                       int accum_dx, int accum_dw, float drop_p, uint32_t drop_seed,
                       int in_group, int in_group_stride, void* dx_bf16, float dx_drop_p, uint32_t dx_drop_seed,
                       void* stream);
/* same, with dy handed over as `nslabs` split-K slabs of the producing GEMM (vlt5_gemm_desc.defer_reduce): row r of dy =
 * sum_s dy[s*slab_stride + r*d ..] in slab order -- saves the separate slab reduction launch and a round trip of dy */
int vlt5_layernorm_bwd_slabs(const float* dy, int nslabs, long long slab_stride, const float* x, const float* w,
                             const float* rstd, float* dx, float* dw, float* dw_partial, int rows, int d,
                             int accum_dx, int accum_dw, float drop_p, uint32_t drop_seed,
                             int in_group, int in_group_stride, void* dx_bf16, float dx_drop_p, uint32_t dx_drop_seed,
                             void* stream);
/* same, plus xn_out_bf16 (optional): the norm's FORWARD output bf16(x * rstd * w) is written there -- the operand of the weight
 * gradient of the projection behind the norm, for a forward that folded the norm around its GEMMs and never materialised it */
int vlt5_layernorm_bwd_full(const float* dy, int nslabs, long long slab_stride, const float* x, const float* w,
                            const float* rstd, float* dx, float* dw, float* dw_partial, int rows, int d,
                            int accum_dx, int accum_dw, float drop_p, uint32_t drop_seed,
                            int in_group, int in_group_stride, void* dx_bf16, float dx_drop_p, uint32_t dx_drop_seed,
                            void* xn_out_bf16, void* stream);
/* job j < njobs (<= 64): out_base[out_off[j] + c] = sum_{b < nblk[j]} partial[(j*slot_rows + b)*width + c], c < width.
 * out_off / nblk are HOST arrays (passed by value to the kernel). */
int vlt5_colsum_multi(const float* partial, float* out_base, const long long* out_off, const int* nblk, int njobs,
                      int slot_rows, int width, void* stream);
int vlt5_layernorm_bwd_blocks(int rows);

/* ---- attention core: softmax(q k^T + bias + masks) v ------------------------------------------
 * replaces the matmul/softmax/dropout/matmul of HF T5Attention.forward (NO 1/sqrt(d) scaling;
 * softmax in f32) for the encoder self-attention (src/modeling_t5_our.py:258-293), the decoder
 * causal self-attention and the decoder cross-attention over the 58 encoder tokens (:641-655).
 * One workgroup per (batch, head); Tq, Tk <= 64, dk <= 64. */
typedef struct {
    const void *q, *k, *v;                   /* bf16; head h of token t of sample b at b*sb + t*st + h*dk */
    long long q_sb, q_st, k_sb, k_st, v_sb, v_st;
    void* ctx; long long o_sb, o_st;         /* bf16 out (fwd) */
    float* lse;                              /* f32 [B,H,Tq]: row max + log row sum (saved for bwd) */
    const float* bias; int bias_q, bias_k;   /* f32 [H,bias_q,bias_k] added where i<bias_q && j<bias_k; NULL = none */
    const float* key_mask; float mask_value; /* f32 [B,Tk] 1=keep 0=pad -> adds (1-m)*mask_value (-1e4 / -1e9) */
    int causal;                              /* adds -1e4 where j > i (HF 4.2.1 extended causal mask) */
    int B, H, Tq, Tk, dk;
    float drop_p; uint32_t drop_seed;        /* dropout on the probabilities */
    /* backward only */
    const void* d_ctx; long long do_sb, do_st;
    void *dq, *dk_, *dv; long long dq_sb, dq_st, dk_sb, dk_st, dv_sb, dv_st;
    float* dbias;                            /* f32 [B,H,bias_q,bias_k] (per-sample dS block) or NULL */
    int fused_heads;                         /* vlt5_qkv_attn_fwd* only: heads per workgroup, 0 = default (vlt5_tuning.fused_heads) */
} vlt5_attn_desc;
int vlt5_attn_fwd(const vlt5_attn_desc* d, void* stream);
int vlt5_attn_bwd(const vlt5_attn_desc* d, void* stream);

/* ---- fused encoder self-attention (north star "fused attention kernel") --------------------------------------------
 * vlt5_qkv_attn_fwd: ONE kernel for the q|k|v projection of LN(x) and the attention core (csrc/enc_attn.hip): a workgroup takes
 * two samples x two heads (128 rows x 384 projection columns on MFMA, then the four 64x64 cores out of LDS), q/k/v only go to
 * HBM once for the backward.  Replaces nn.Linear q/k/v + matmul/softmax/dropout/matmul of HF T5Attention.forward for the encoder
 * (VL-T5/src/modeling_t5_our.py:282-293).  d_kv = 64, H even, S <= 64, d_model % 64 == 0; `core` as for vlt5_attn_fwd with
 * q/k/v = qkv_bf16 + {0, H*64, 2*H*64}, token stride 3*H*64, sample stride S*3*H*64, Tq = Tk = S.  Bit-identical to
 * vlt5_gemm_bf16 + vlt5_attn_fwd. */
int vlt5_qkv_attn_fwd(const void* xn_bf16, const void* wqkv_bf16, void* qkv_bf16, const vlt5_attn_desc* core, int d_model, void* stream);
/* the same kernel with the T5 RMS norm folded in (HF T5LayerSelfAttention: T5LayerNorm -> q/k/v): xw_bf16 = bf16(x * w_norm) and the
 * per-row partial sums of squares of x as the producing GEMM's epilogue left them (vlt5_gemm_desc.emit_xw_bf16 / emit_partials);
 * the q|k|v rows are scaled by rstd[m] = rsqrt(sum of the norm_nparts (<= 32) partials / d_model + norm_eps) on their way out of
 * the accumulators, rstd goes to norm_rstd_out [B*S] (optional) for the norm's backward. */
int vlt5_qkv_attn_fwd_norm(const void* xw_bf16, const void* wqkv_bf16, void* qkv_bf16, const vlt5_attn_desc* core, int d_model,
                           const float* norm_partials, int norm_nparts, float norm_eps, float* norm_rstd_out, void* stream);

/* Fused decoder attention sublayers (between the two norms): projection of one head + attention core + that head's share of the
 * output projection, one workgroup per (sample group, head) -- three launches of a 400-row decoder sublayer in one.
 * Replaces HF T5LayerSelfAttention / T5LayerCrossAttention.forward without the norm and the residual (VL-T5/src/modeling_t5_our.py:641-655
 * -> HF T5Block): the H output slabs are summed -- with dropout and the residual -- by vlt5_layernorm_fwd_slabs.
 * core: as for vlt5_attn_fwd (d_kv = 64, Tq <= 16, Tk <= 64); proj_bf16, core.ctx and core.lse are written for the backward.
 *   self : w_bf16 [3*H*64, d_model] q | k | v rows; proj_bf16 [B*Tq, 3*H*64]; core.q / k / v and their strides must describe proj_bf16
 *   cross: w_bf16 [H*64, d_model] q rows; proj_bf16 [B*Tq, H*64] (= core.q); core.k / core.v: the projected encoder-side keys / values */
typedef struct vlt5_dec_attn_desc {
    const void* xn_bf16;      /* [B*Tq, d_model] bf16: the normalised sublayer input */
    const void* w_bf16;
    const void* wo_bf16;      /* [d_model, H*64] bf16 output projection */
    void* proj_bf16;
    float* o_slabs;           /* out: H slabs of [B*Tq, d_model] f32; slab h = ctx_h . Wo[:, h*64 .. h*64+63]^T */
    long long slab_stride;    /* elements between slabs (>= B*Tq*d_model, multiple of 4) */
    int d_model;              /* multiple of 64 */
    vlt5_attn_desc core;
} vlt5_dec_attn_desc;
int vlt5_dec_self_attn_fwd(const vlt5_dec_attn_desc* d, void* stream);
int vlt5_cross_attn_fwd(const vlt5_dec_attn_desc* d, void* stream);
/* 1 if both kernels serve answers of T tokens against Tk_cross encoder-side keys at this head / model width */
int vlt5_dec_attn_fused_ok(int T, int Tk_cross, int d_kv, int d_model);
/* the whole sublayer: x_out = x + dropout(o(attention(LN(x)))) -- HF T5LayerSelfAttention.forward; norm + fused kernel + output
 * projection (dropout + residual in its epilogue); xn / rstd / qkv / ctx / lse are left for the backward */
typedef struct {
    const float* x; const float* ln_w;               /* f32 [B*S, d_model] residual stream in, norm weight [d_model] */
    const void* wqkv_bf16; const void* wo_bf16;      /* bf16 [3*H*64, d_model] (q|k|v rows), [d_model, H*64] */
    float* x_out;                                    /* f32 [B*S, d_model] */
    void* xn_bf16; float* rstd; void* qkv_bf16; void* ctx_bf16; float* lse;   /* saved: [B*S,d], [B*S], [B*S,3*H*64], [B*S,H*64], [B,H,S] */
    const float* bias; int bias_q, bias_k;           /* f32 [H,bias_q,bias_k] relative-position bias block, or NULL */
    const float* key_mask; float mask_value;         /* f32 [B,S] 1 = keep, adds (1-m)*mask_value */
    int B, S, H, d_model; float eps;
    float drop_p; uint32_t seed_probs, seed_out;     /* dropout on the probabilities / on the sublayer output */
} vlt5_enc_attn_desc;
int vlt5_enc_attn_fwd(const vlt5_enc_attn_desc* d, void* stream);
/* backward of the sublayer from what the forward saved (autograd's backward of the same modules, src/vqacl.py:461).  dx may alias dy.
 * d_scores (optional): f32 [B,H,bias_q,bias_k] per-sample gradient of the bias block (reduce with vlt5_relbias_bwd).
 * workspace: vlt5_enc_attn_bwd_workspace_bytes(...) bytes of device scratch. */
typedef struct {
    const float* dy;                                 /* f32 [B*S, d_model] gradient of x_out */
    float* dx;                                       /* f32 [B*S, d_model] gradient of x */
    float* d_wqkv; float* d_wo; float* d_ln_w;       /* f32 [3*H*64, d_model], [d_model, H*64], [d_model] (overwritten) */
    float* d_scores;
} vlt5_enc_attn_grads;
long long vlt5_enc_attn_bwd_workspace_bytes(int B, int S, int H, int d_model);
int vlt5_enc_attn_bwd(const vlt5_enc_attn_desc* d, const vlt5_enc_attn_grads* g, void* workspace, void* stream);

/* backward of the two decoder attention sublayers above (autograd's backward of HF T5Attention inside T5LayerSelfAttention /
 * T5LayerCrossAttention, src/vqacl.py:461), from what the forward left: xn_bf16, proj_bf16, core.ctx, core.lse (core.k / core.v for the
 * cross sublayer).  d_out f32 [B*Tq, d_model]: gradient of the sublayer output in front of the residual (= of the sum of the H slabs;
 * the output dropout belongs to the caller, like the slab sum).  Outputs: d_xn f32 [B*Tq, d_model] (gradient of the normalised input),
 * d_w f32 (shape of w_bf16), d_wo f32 [d_model, H*d_kv], d_proj bf16 (shape of proj_bf16: the gradient of q | k | v, or of q), cross: dk / dv
 * bf16 gradients of the encoder-side keys / values with sample / token strides dkv_sb / dkv_st; d_scores (optional) as for vlt5_attn_bwd.
 * workspace: vlt5_dec_attn_bwd_workspace_bytes(...) bytes. */
typedef struct {
    const float* d_out; float* d_xn; float* d_w; float* d_wo; void* d_proj;
    void* dk; void* dv; long long dkv_sb, dkv_st;
    float* d_scores;
} vlt5_dec_attn_grads;
long long vlt5_dec_attn_bwd_workspace_bytes(int B, int Tq, int H, int d_kv, int d_model);
int vlt5_dec_self_attn_bwd(const vlt5_dec_attn_desc* d, const vlt5_dec_attn_grads* g, void* workspace, void* stream);
int vlt5_cross_attn_bwd(const vlt5_dec_attn_desc* d, const vlt5_dec_attn_grads* g, void* workspace, void* stream);

/* ---- feed-forward sublayer: x_out = x + dropout(wo(dropout(act(wi(LN(x)))))) -- HF T5LayerFF.forward with T5DenseReluDense (ReLU) or,
 * gated = 1, T5DenseGatedGeluDense (gelu_new(wi_0 x) * wi_1 x; wi_bf16 = [wi_0; wi_1] rows), as called from the encoder / decoder T5Blocks
 * (src/modeling_t5_our.py:282-293, 641-655).  xn / rstd / h (and u for the gated form) are left for the backward. */
typedef struct {
    const float* x; const float* ln_w;            /* f32 [M, d_model], [d_model] */
    const void* wi_bf16; const void* wo_bf16;     /* bf16 [d_ff (gated: 2*d_ff), d_model], [d_model, d_ff] */
    float* x_out;                                 /* f32 [M, d_model] */
    void* xn_bf16; float* rstd; void* h_bf16; void* u_bf16;     /* saved: [M,d_model], [M], [M,d_ff], gated: [M,2*d_ff] */
    int M, d_model, d_ff, gated; float eps;
    float drop_p; uint32_t seed_hidden, seed_out; /* dropout on the hidden activation / on the sublayer output */
    const vlt5_tuning* tuning;
} vlt5_ffn_desc;
typedef struct { const float* dy; float* dx; float* d_wi; float* d_wo; float* d_ln_w; } vlt5_ffn_grads;   /* dx may alias dy; weight grads overwritten */
int vlt5_ffn_fwd(const vlt5_ffn_desc* d, void* stream);
long long vlt5_ffn_bwd_workspace_bytes(int M, int d_model, int d_ff, int gated);
int vlt5_ffn_bwd(const vlt5_ffn_desc* d, const vlt5_ffn_grads* g, void* workspace, void* stream);

/* ---- rescale + tied lm_head + cross-entropy (src/modeling_t5_our.py:661-686): logits = (x * d_model^-0.5) E^T,
 * loss_tok = CrossEntropyLoss(ignore_index=-100, reduction='none'); backward: dlogits (bf16 scratch [rows, vocab]), d_x f32 [rows, d_model],
 * d_emb f32 [vocab, d_model] (accum_d_emb = 1: added to, the embedding gathers contribute to the same tied tensor). */
typedef struct {
    const void* x_bf16; const void* emb_bf16;     /* bf16 [rows, d_model] decoder output, [vocab, d_model] shared embedding */
    const long long* labels;                      /* [rows], -100 = ignore */
    float* logits; float* loss_tok; float* lse;   /* out: f32 [rows, vocab], [rows], [rows] */
    int rows, d_model, vocab;
    const vlt5_tuning* tuning;
} vlt5_lmhead_ce_desc;
typedef struct { const float* d_loss_tok; void* dlogits_bf16; float* d_x; float* d_emb; int accum_d_emb; } vlt5_lmhead_ce_grads;
int vlt5_lmhead_ce_fwd(const vlt5_lmhead_ce_desc* d, void* stream);
int vlt5_lmhead_ce_bwd(const vlt5_lmhead_ce_desc* d, const vlt5_lmhead_ce_grads* g, void* stream);

/* ---- relative position bias: HF T5Attention.compute_bias / _relative_position_bucket ---------
 * The integer bucket table lut[Lq*Lk] is computed on the host (vqacl_amd/buckets.py, bit-exact
 * with the library); the kernel gathers table[lut[i,j], h] into bias[h,i,j]. */
int vlt5_relbias_build(const float* table, const int* lut, float* bias, int H, int Lq, int Lk, int nbuckets, void* stream);
/* dtable[bucket,h] (+)= sum over nmat matrices and positions of dS[mat,h,i,j];  scratch f32 [64*H*Lq*Lk] */
int vlt5_relbias_bwd(const float* dS, const int* lut, float* dtable, float* scratch, int nmat, int H, int Lq, int Lk,
                     int nbuckets, int accum, void* stream);

/* ---- embeddings --------------------------------------------------------------------------------
 * token gather: embed_tokens(input_ids) (src/modeling_t5_our.py:196) / decoder embed of the
 * shifted labels, plus the stack's input dropout (:247).  Row (b,t) goes to out + b*out_sb + t*out_st;
 * the dropout index is ((b*drop_rows + drop_row0 + t)*d + c). */
int vlt5_embed_fwd(const long long* ids, const float* table, float* out, long long out_sb, long long out_st,
                   int B, int T, int d, int vocab, float drop_p, uint32_t drop_seed, int drop_rows, int drop_row0, void* stream);
/* dtable[ids[b,t]] += dropout'(dout[b,t]) -- the scatter-add of nn.Embedding's backward (src/vqacl.py:461), DETERMINISTIC: the
 * contributions of an id that occurs several times are summed in a fixed two-level tree, no atomics.  scratch:
 * vlt5_embed_bwd_scratch_bytes(B, T, d) bytes of device memory (16-byte aligned). */
long long vlt5_embed_bwd_scratch_bytes(int B, int T, int d);
int vlt5_embed_bwd(const long long* ids, const float* dout, long long sb, long long st, float* dtable,
                   int B, int T, int d, int vocab, float drop_p, uint32_t drop_seed, int drop_rows, int drop_row0, void* scratch,
                   void* stream);
/* bf16 staging mirror of a scatter-added gradient table (the data-parallel bf16 bucket that holds `shared.weight`: its dense part is
 * mirrored by the lm_head weight-gradient GEMM, vlt5_gemm_desc.c_bf16_copy; the rows the embedding backward added to afterwards are
 * re-rounded here): rows ids0[0..n0), ids1[0..n1) (out-of-range ids clamp to row 0 like the lookups) and the last `tail_rows` rows
 * (the object-order rows of VisualEmbedding, src/modeling_t5_our.py:126-133) of src f32 [vocab, d] -> dst_bf16, same offsets */
int vlt5_mirror_rows_bf16(const float* src, void* dst_bf16, int vocab, int d, const long long* ids0, int n0, const long long* ids1,
                          int n1, int tail_rows, void* stream);
/* labels -> decoder input ids (HF _shift_right, called at src/modeling_t5_our.py:620) */
int vlt5_shift_right(const long long* labels, long long* out, int B, int T, int start_id, int pad_id, void* stream);
/* f32 [B,S] encoder mask: 1 where input_ids != pad for the L text columns, 1 for the rest (:225-232, :631-638) */
int vlt5_build_mask(const long long* ids, float* mask, int B, int L, int S, int pad_id, void* stream);

/* the inputs of one T5 stack in ONE launch: vlt5_build_mask(mask_ids [B,L] -> mask [B,S]) + vlt5_relbias_build + vlt5_embed_fwd of
 * `ids` [B,T] -- or, with `labels` != NULL (the decoder, src/modeling_t5_our.py:620), of _shift_right(labels), whose ids are also
 * stored to ids_out [B,T] for the backward's scatter.  Element for element the arithmetic of the four kernels. */
typedef struct {
    const long long* mask_ids; float* mask; int B, L, S;
    const float* rel_table; const int* lut; float* bias; int H, Lq, Lk;
    const long long* ids; const long long* labels; long long* ids_out; int T, start_id, pad_id;
    const float* table; float* out; long long out_sb, out_st; int d, vocab; float drop_p; uint32_t drop_seed; int drop_rows, drop_row0;
} vlt5_stack_inputs_desc;
int vlt5_stack_inputs_fwd(const vlt5_stack_inputs_desc* s, void* stream);

/* ---- VisualEmbedding.forward (src/modeling_t5_our.py:93-143) after the 2048->d projection ------
 * out[b, row0+i] = drop( LN(G[b,i]) + LN(Wp [box,area] + bp) + img_order[0] + shared[vocab-1-i] )
 * "area" reads the box columns as (x1,x2,y1,y2) exactly like get_area (:78-90). */
int vlt5_vis_embed_fwd(const float* G, const float* boxes, const float* Wp, const float* bp, const float* lnf_w,
                       const float* lnp_w, const float* img0, const float* shared, float* out, long long out_sb,
                       long long out_st, float* rstd_f, float* rstd_p, int B, int V, int d, int vocab, float eps,
                       float drop_p, uint32_t drop_seed, int drop_rows, int drop_row0, void* stream);
/* dG bf16 [B*V,d] (input of the projection's wgrad GEMM); parameter grads via per-split partials:
 * partial f32 [nsplit = vlt5_vis_embed_bwd_blocks(B*V)][10*d] laid out per split as
 * [dlnf_w d | dlnp_w d | dbp d | dWp d*5 (row-major [d][5]) | dimg0 d | dbf d], followed by 2*B*V floats of scratch;
 * reduce with vlt5_colsum(partial, out, nsplit, 10*d).  dshared rows vocab-1-i are accumulated (+=) in a fixed order. */
int vlt5_vis_embed_bwd(const float* dout, long long sb, long long st, const float* G, const float* boxes, const float* Wp,
                       const float* bp, const float* lnf_w, const float* lnp_w, const float* rstd_f, const float* rstd_p,
                       void* dG_bf16, float* partial, float* dshared, int B, int V, int d, int vocab,
                       float drop_p, uint32_t drop_seed, int drop_rows, int drop_row0, void* stream);
int vlt5_vis_embed_bwd_blocks(int rows);
/* reduced = column sums of the partial rows (vlt5_colsum over all 10*d columns) -> the six visual-embedding parameter
 * gradients: feat LN weight [d], pos LN weight [d], pos bias [d], pos weight [d,5], img_order_embedding [n_images,d]
 * (row 0 only, others zero), feat bias [d]. */
int vlt5_vis_grad_scatter(const float* reduced, float* g_lnf, float* g_lnp, float* g_bp, float* g_wp, float* g_img,
                          float* g_bf, int d, int n_images, void* stream);
/* out[c] (+)= sum_blk partial[blk*row_stride + c], c < width  (fixed summation order) */
int vlt5_colsum(const float* partial, float* out, int nblk, int width, int row_stride, int accum, void* stream);

/* ---- lm_head cross-entropy (src/modeling_t5_our.py:680-686) and the train_step reduction
 *      (src/vqa_model.py:46-54) ----------------------------------------------------------------- */
int vlt5_ce_fwd(const float* logits, const long long* labels, float* loss_tok, float* lse, int R, int V, void* stream);
/* loss = mean_b( score_b * sum_t CE_bt m_bt / max(sum_t m_bt,1) );  row_w[b,t] = d loss / d CE_bt */
int vlt5_loss_reduce(const float* loss_tok, const long long* labels, const float* scores, float* loss, float* row_w,
                     int B, int T, void* stream);
/* out[r] = index of the first maximum of row r of x [rows, cols] (greedy decoding over the vocabulary) */
int vlt5_argmax_rows(const float* x, int rows, int cols, long long* out, void* stream);
/* dlogits[r,:] = (softmax(logits[r]) - onehot(label_r)) * row_w[r] * (*gout or 1), bf16, 0 for ignored rows */
int vlt5_ce_bwd(const float* logits, const long long* labels, const float* lse, const float* row_w, const float* gout,
                void* dlogits_bf16, int R, int V, void* stream);

/* ---- SS/SI prototype head (src/modeling_t5_our.py:434-511, :583-615) -------------------------- */
/* mean over tokens [0,min(split,S)) -> poolQ, [split,S) -> poolV   (hidden f32, sample b at hidden + b*sb, rows of d) */
int vlt5_proto_pool(const float* hidden, long long sb, int B, int S, int d, int split, float* poolQ, float* poolV, void* stream);
/* calculate_current_prototype: proto[c] = sum_b onehot[b,c] pool[b] / max(cnt_c,1), cnt = sum_b onehot */
int vlt5_proto_class_mean(const float* pool, const float* onehot, float* proto, float* cnt, int B, int C, int d, void* stream);
/* data parallel (vqacl_amd/prototype.py::_allreduce_stats): the class statistics of both heads as ONE buffer for a single all-reduce,
 * packed = [curQ * max(numQ,1) | numQ | curV * max(numV,1) | numV] ((CQ + CV) * (d + 1) floats); unpack != 0: the way back, class
 * means over the global batch = summed sums / max(summed counts, 1), counts = summed counts */
int vlt5_proto_stats_pack(float* curQ, float* numQ, float* curV, float* numV, float* packed, int CQ, int CV, int d, int unpack,
                          void* stream);
/* update_prototype state machine, all branches; see vqacl_amd/prototype.py for the host side.
 * first: 1 on the first batch of `task`.  qmem: this task's memory tensor [CQ,d] (NULL when task==0). */
int vlt5_proto_update(const float* curQ, const float* curV, const float* numQ, const float* numV, float* Qproto,
                      float* Vproto, float* Qnum, float* Vnum, float* qmem, int qmem_initialised, int first, int task,
                      float alpha, float beta, int CQ, int CV, int d, void* stream);
/* cosine_similarity_multi + gather: idx[b] = argmax_c cos(tanh P_c, tanh pool_b) (first max wins);
 * the selected row is written as f32 (out_f32 + b*sb, may be NULL) and bf16 (out_bf16 + b*sb_bf16, may be NULL) */
int vlt5_proto_retrieve(const float* protos, const float* pool, long long* idx, float* out_f32, long long sb,
                        void* out_bf16, long long sb_bf16, float* scratch /* f32 [C*d] */, int B, int C, int d, void* stream);
/* memory_loss (nextqa/modeling_t5_nextqa.py:544-555): out[0] = mean_b ||pool_b - (onehot P)_b||^2 */
int vlt5_proto_memory_loss(const float* pool, const float* onehot, const float* protos, float* out, int B, int C, int d, void* stream);

/* The whole head of one forward in three launches (token pooling; one workgroup per prototype row: class mean of the batch, per-task
 * state update, tanh-normalised copy; retrieval of both heads) -- the same arithmetic as vlt5_proto_pool / _class_mean / _update /
 * _retrieve, bit for bit.  hidden f32 [B, >= S, d] (sample stride hidden_sb): the encoder output; rows [0, split) pool to Q, [split, S)
 * to V.  update = 1: calculate_current_prototype + update_prototype with the one-hot labels (first / task / qmem* as for
 * vlt5_proto_update; the per-task control flow stays with the caller); update = 0: retrieval from the current prototypes only.
 * The retrieved Q / V prototype of sample b is written to out_f32 + b*out_sb (+ d for V) and out_bf16 + b*out_sb_bf16 (+ d) -- rows
 * S, S+1 of the decoder's memory -- either may be NULL.  scratch: f32 [(CQ + CV) * d]. */
typedef struct {
    const float* hidden; long long hidden_sb; int B, S, d, split;
    float *poolQ, *poolV;                       /* out: [B, d] each */
    const float *onehotQ, *onehotV;             /* [B, CQ], [B, CV] (update only) */
    float *Qproto, *Vproto, *Qnum, *Vnum;       /* state: [CQ, d], [CV, d], [CQ], [CV] */
    float* qmem; int qmem_initialised, first, task, update;
    float alpha, beta;
    int CQ, CV;
    long long *idxQ, *idxV;                     /* out: [B] argmax indices */
    float* out_f32; long long out_sb; void* out_bf16; long long out_sb_bf16;
    float* scratch;
    /* data parallel (round 5): the head in two halves around ONE all-reduce of `packed` -- [class sums Q (CQ*d) | counts Q (CQ) | class sums
     * V (CV*d) | counts V (CV)], (CQ + CV) * (d + 1) floats.  phase 1: pooling + the batch's class SUMS and counts into `packed` (no state
     * change, no retrieval); phase 2: class means of the GLOBAL batch = packed sums / max(count, 1), the state update, the normalised
     * copies and the retrieval of both heads.  phase 0 (default): the single-process head, `packed` unused. */
    float* packed; int phase;
} vlt5_proto_head_desc;
int vlt5_proto_head_fwd(const vlt5_proto_head_desc* h, void* stream);

/* ---- optimizer: clip_grad_norm_(5) + HF AdamW (src/vqacl.py:466-487, src/trainer_base.py:187-190) */
/* partial: vlt5_sqnorm_blocks(n) floats.  accum_total: 0 total_sq = sum, 1 total_sq += sum, 2 leave only the block partials (no
 * reduction: several ranges are then summed by one vlt5_gnorm_finish over the concatenated partials) */
int vlt5_sqnorm(const float* g, long long n, float* partial, float* total_sq, int accum_total, void* stream);
/* total_sq[0] = sum(partials[0..nslots)) + sum over the `nranges` (<= 4) element ranges [range_off[i], +range_n[i]) of grads of g^2,
 * all in a fixed order (deterministic).  `scratch`: vlt5_sqnorm_blocks(sum of range_n) + nranges floats.  Completes the
 * gradient norm whose per-tile shares the weight-gradient GEMMs left in vlt5_step.gnorm_partials (reference: the norm pass of
 * torch.nn.utils.clip_grad_norm_, src/vqacl.py:466-487). */
int vlt5_gnorm_finish(const float* partials, long long nslots, const float* grads, const long long* range_off, const long long* range_n,
                      int nranges, float* scratch, float* total_sq, void* stream);
int vlt5_sqnorm_blocks(long long n);
/* p,m,v f32; optional bf16 shadow of p.  clip coefficient = min(1, max_norm / (sqrt(*total_sq) + 1e-6)) when
 * total_sq != NULL.  hf_mode 1: transformers AdamW (eps outside bias correction, decay after the update, on the
 * updated value); 0: torch.optim.AdamW ordering. */
int vlt5_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, long long n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, const float* total_sq, float max_norm,
                    int hf_mode, void* stream);
/* the same two with the gradient read as bf16 * g_scale -- the reduced bucket of a bf16 data-parallel all-reduce as it lies in
 * the staging buffer (g_scale = 1/world): what vlt5_cast_f32 would have written to the f32 gradient buffer, without the pass */
int vlt5_sqnorm_g16(const void* g_bf16, float g_scale, long long n, float* partial, float* total_sq, int accum_total, void* stream);
int vlt5_adamw_step_g16(float* p, const void* g_bf16, float g_scale, float* m, float* v, void* p_bf16, long long n, float lr,
                        float beta1, float beta2, float eps, float weight_decay, int step, const float* total_sq, float max_norm,
                        int hf_mode, void* stream);
int vlt5_cast_bf16(const float* src, void* dst_bf16, long long n, void* stream);
/* dst = scale * float(src_bf16): the way back from a bf16 gradient all-reduce (vqacl_amd/parallel.py) */
int vlt5_cast_f32(const void* src_bf16, float* dst, long long n, float scale, void* stream);
int vlt5_scale_add(float* dst, const float* src, float a, float b, long long n, void* stream); /* dst = a*dst + b*src */

/* ---- gated-GELU FFN activation (HF T5DenseGatedActDense.forward: gelu_new(wi_0 x) * wi_1 x, dropout) --------------------
 * u bf16 [rows, 2*ff] = x [wi_0; wi_1]^T from one GEMM; h bf16 [rows, ff] = dropout(gelu_new(u[:, :ff]) * u[:, ff:]),
 * dropout element index r*ff+c.  bwd: du[:, :ff] = dh' u1 gelu_new'(u0), du[:, ff:] = dh' gelu_new(u0), dh' = dropout-backward(dh). */
int vlt5_glu_fwd(const void* u_bf16, void* h_bf16, long long rows, int ff, float drop_p, uint32_t drop_seed, void* stream);
int vlt5_glu_bwd(const void* dh_bf16, const void* u_bf16, void* du_bf16, long long rows, int ff, float drop_p, uint32_t drop_seed,
                 void* stream);

/* dst_bf16 = dropout(src) (inverted, element index r*cols+c) cast to bf16; the bf16 operand of the backward GEMMs */
int vlt5_drop_cast(const float* src, void* dst_bf16, long long rows, int cols, float drop_p, uint32_t drop_seed, void* stream);

/* ================================================================================================
 * Whole-path engine: VLT5.forward / backward (src/modeling_t5_our.py:514-713, src/vqacl.py:461)
 * composed from the kernels above, one C call per phase so the host does O(1) work per step.
 *
 *   vlt5_encoder_fwd   JointEncoder.forward (:175-339)            -> enc_out f32 [B,S,d], enc_ext bf16 rows 0..S-1
 *   (host: SS/SI prototype head, vqacl_amd/prototype.py, writes rows S, S+1 of enc_ext)
 *   vlt5_decoder_fwd   shift-right, decoder T5Stack (:641-655), rescale + lm_head + CE (:661-686),
 *                      train_step loss reduction (src/vqa_model.py:46-54)
 *   vlt5_decoder_bwd / vlt5_encoder_bwd   the autograd backward of the above into the flat grad buffer
 *
 * Parameters live in ONE flat f32 buffer (+ a bf16 shadow with identical offsets, + a flat f32 gradient
 * buffer); vlt5_layout_* describes it with the reference's state_dict names, ordered so that gradients
 * complete front-to-back during backward (contiguous all-reduce buckets for data parallelism).
 * ================================================================================================ */
typedef struct {
    int d_model, d_kv, num_heads, d_ff, num_layers, num_decoder_layers, vocab;
    int rel_buckets, feat_dim, n_images;
    int pad_id, dec_start_id;
    int n_ques, n_cate;            /* prototype classes (10 question types, 80 categories) */
    float eps, dropout;
    int gated_act;                 /* 0: ReLU FFN (t5-base / t5-large: every reference configuration); 1: gated GELU (HF
                                      T5DenseGatedActDense, feed_forward_proj = "gated-gelu"): wi_0 | wi_1 adjacent in the layout */
} vlt5_config;

typedef struct {
    int B, L, V, T;                /* batch, text tokens, visual tokens, answer tokens */
    int training;                  /* 1: dropout active */
    uint32_t seed;                 /* dropout base seed of this step */
    const float* params;           /* flat f32 master parameters */
    const void* params_bf16;       /* flat bf16 shadow, same element offsets */
    float* grads;                  /* flat f32 gradients (backward only) */
    void* workspace; long long workspace_bytes;
    const float* vis_feats;        /* [B,V,feat_dim] */
    const float* boxes;            /* [B,V,4] */
    const long long* input_ids;    /* [B,L] */
    const long long* labels;       /* [B,T], -100 = ignore */
    const float* scores;           /* [B] answer scores, or NULL: skip the fused train_step reduction */
    const int* enc_lut;            /* [L*L] bidirectional bucket ids */
    const int* dec_lut;            /* [T*T] causal bucket ids */
    const float* gout;             /* [1] upstream gradient of the reduced loss or NULL (=1); used when d_loss_tok == NULL */
    const float* d_loss_tok;       /* [B*T] upstream gradient of the per-token loss (generic VLT5.forward path) or NULL */
    void** events; int n_events;   /* optional hipEvent_t recorded when a gradient bucket is complete (backward) */
    void** wait_events; int n_wait_events;   /* optional hipEvent_t per parameter bucket (same numbering): the forward phases
                                      make the stream wait for bucket b's event before the first kernel that reads that
                                      bucket's weights -- lets an optimizer update on another stream overlap the forward */
    /* optional HBM-resident region-feature store (vlt5_feat_store_put): when feat_store != NULL the step's visual inputs are
     * rows of the store selected by feat_slots and vis_feats / boxes are ignored (may be NULL) */
    const void* feat_store;        /* bf16 [n_slots][V][feat_dim] */
    const float* box_store;        /* f32  [n_slots][V][4], normalised boxes */
    const long long* feat_slots;   /* [B] slot of each sample */
    long long n_slots;
    /* optional second stream for the backward phases: the batched weight-gradient GEMMs are enqueued there (after an event of the
     * main stream) and run beside the input-gradient chain; vlt5_encoder_bwd makes the main stream wait for it before it
     * returns, so the caller sees a single-stream contract.  side_events: >= 4 hipEvent_t owned by the caller.  NULL: one stream. */
    void* side_stream; void** side_events; int n_side_events;
    /* optional bf16 mirror of `grads` (same element offsets): the backward phases write every weight gradient that a GEMM
     * produces, and the relative-position tables, there as well, so a bf16 data-parallel all-reduce of the layer buckets needs no
     * cast pass.  NOT covered (cast them): the last bucket (embeddings, norms, visual embedding). */
    void* grads_bf16;
    /* optional: vlt5_gnorm_slots(config) floats.  The backward phases zero it (vlt5_decoder_bwd) and every weight-gradient GEMM of the
     * layer buckets and of the stacked cross-attention K/V projection leaves the sum of squares of its output tiles there
     * (vlt5_gemm_desc.sumsq; slot = ceil(flat offset / 4096) + tile) -- with vlt5_gnorm_finish over the remaining ranges (last
     * bucket, relative-position tables) the optimizer's global gradient norm needs no second pass over 0.9 GB of gradients. */
    float* gnorm_partials;
    /* 1: vlt5_decoder_bwd leaves five of the decoder's six batched weight gradients to vlt5_encoder_bwd, which launches them as the
     * second problem of its long weight-gradient launches (they run on the CUs those leave idle).  The decoder's weight gradients
     * are then complete only after vlt5_encoder_bwd; ignored with a side stream / gradient-bucket events (data parallelism). */
    int defer_decoder_wgrads;
    const vlt5_tuning* tuning;     /* optional experiment switches (NULL = defaults) */
    /* optional, with `events`: the id vlt5_grad_release_plan() returned when the caller cut its event waits / collectives.  The backward
     * phases recompute it from what THEY are about to do (configuration, tuning, side stream, defer_decoder_wgrads) and return
     * VLT5_ERR_PLAN before launching anything when it differs -- a caller that waits for a bucket's event after the wrong call would
     * otherwise read the PREVIOUS backward's record without an error.  0: not checked. */
    int release_plan_id;
} vlt5_step;
/* number of slots of vlt5_step.gnorm_partials for this configuration, or 0 when it is not supported (a matrix dimension that is
 * no multiple of 64: the slot ranges of neighbouring tensors would overlap) */
long long vlt5_gnorm_slots(const vlt5_config* c);

/* ---- batch feed from a resident feature store (replaces the per-item HDF5 read + collate + H2D copy of
 *      src/vqa_data_memory.py:141-189, 291-396) ----
 * put: store[slots[i]] = bf16(feats[i]) (round to nearest even: exactly the rounding the engine applies to the f32 batch),
 *      box_store[slots[i]] = boxes[i]; feats f32 [n,V,feat_dim], boxes f32 [n,V,4] device pointers.
 * gather: out_feats[b] = store[slots[b]], out_boxes[b] = box_store[slots[b]]; a slot outside [0,n_slots) yields a zero row
 *      (put: is skipped) -- the host side validates before launching. feat_dim % 8 == 0, V <= 256. */
int vlt5_feat_store_put(const float* feats, const float* boxes, const long long* slots, int n, void* store_bf16,
                        float* box_store, long long n_slots, int V, int feat_dim, void* stream);
int vlt5_feat_gather(const void* store_bf16, const float* box_store, const long long* slots, long long n_slots,
                     void* out_feats_bf16, float* out_boxes, int B, int V, int feat_dim, void* stream);

/* vlt5_encoder_bwd completes the weight gradients of the encoder in two groups (events of vlt5_step.events): layers
 * [late, num_layers) mid-phase, layers [0, late) at the end; returns `late` */
int vlt5_encoder_late_layers(int num_layers);
int vlt5_encoder_late_layers_tuned(int num_layers, const vlt5_tuning* tuning);   /* with vlt5_tuning.enc_cut */
/* 1: with gradient-bucket events (vlt5_step.events) and vlt5_step.defer_decoder_wgrads = 1 the decoder-layer buckets
 * [0, num_decoder_layers) are signalled from INSIDE vlt5_encoder_bwd, at its mid-point together with the upper half of the encoder
 * (their weight gradients ride in that phase's launches); the caller enqueues its waits for them after that call.  0: signalled at the
 * end of vlt5_decoder_bwd.  The stacked cross-K/V bucket (index num_decoder_layers) is always signalled by vlt5_decoder_bwd. */
int vlt5_decoder_buckets_late(const vlt5_config* c, const vlt5_tuning* tuning, int side_stream);
/* The order a backward WITH gradient-bucket events and defer_decoder_wgrads = 1 completes the buckets in, as the engine itself decides
 * it: up to `cap` triples (phase, lo, hi) into `triples` -- phase 0: buckets [lo, hi) are signalled by vlt5_decoder_bwd, phase 1: by
 * vlt5_encoder_bwd, in this order (a wait for a bucket's event must be enqueued AFTER the call that records it).  Returns the plan's id
 * (> 0, for vlt5_step.release_plan_id; a function of the triples), or a negative value for bad arguments / cap < 5.  *n = triples written. */
int vlt5_grad_release_plan(const vlt5_config* c, const vlt5_tuning* tuning, int side_stream, int* triples, int cap, int* n);
/* a lowest-priority stream for vlt5_step.side_stream (hipStreamCreateWithPriority); the caller destroys it */
int vlt5_side_stream_create(void** stream);
int vlt5_side_stream_destroy(void* stream);

/* parameter layout */
int vlt5_layout_count(const vlt5_config* c);
/* name_cap >= 128.  bucket: index of the gradient bucket (0 = first complete in backward); decay: 1 if the
 * reference's optimizer grouping applies weight decay (src/trainer_base.py:148-161); used: 0 for prototype_fc* */
int vlt5_layout_get(const vlt5_config* c, int i, char* name, int name_cap, long long* offset, int* rows, int* cols,
                    int* bucket, int* decay, int* used);
long long vlt5_layout_total(const vlt5_config* c);
int vlt5_layout_buckets(const vlt5_config* c);

/* workspace */
long long vlt5_workspace_bytes(const vlt5_config* c, int B, int L, int V, int T);
enum { VLT5_WS_ENC_OUT = 0, VLT5_WS_ENC_EXT = 1, VLT5_WS_LOGITS = 2, VLT5_WS_LOSS_TOK = 3, VLT5_WS_LOSS = 4,
       VLT5_WS_ENC_MASK_EXT = 5, VLT5_WS_DEC_OUT = 6 };
long long vlt5_workspace_offset(const vlt5_config* c, int B, int L, int V, int T, int which);   /* byte offset, -1 = unknown */

/* Phase entry points: each enqueues a whole phase of one train / eval step on `stream` from C++ (no host round trips).
 * vlt5_encoder_fwd  = JointEncoder.forward (src/modeling_t5_our.py:175-339): token gather (:196), VisualEmbedding (:93-143),
 *                     relative-position bias + masks (:225-273), the encoder T5Blocks (:282-293), final norm + dropout; leaves the
 *                     encoder output (f32 and bf16, rows 0..S-1 of S+2 per sample) in the workspace for the prototype head.
 * vlt5_decoder_fwd  = the rest of VLT5.forward (:618-713): _shift_right (:620), decoder T5Stack over the S+2 memory rows (:641-655),
 *                     rescale + tied lm_head (:661-671), CrossEntropyLoss(ignore_index=-100, reduction='none') (:680-686) and, when
 *                     s->scores is given, the per-sample reduction of src/vqa_model.py:46-54.
 * vlt5_decoder_bwd / vlt5_encoder_bwd = loss.backward() (src/vqacl.py:461) through the same two halves, gradients into s->grads. */
int vlt5_encoder_fwd(const vlt5_config* c, const vlt5_step* s, void* stream);
int vlt5_decoder_fwd(const vlt5_config* c, const vlt5_step* s, void* stream);
/* One greedy-decoding step with a key/value cache (replaces HF generate -> VLT5.forward(decoder_input_ids[:, -1:],
 * past_key_values), src/modeling_t5_our.py:544-566,624-629,715-772; vqa_model.py:112-116).  s->T is the cache capacity
 * (<= 64), s->training must be 0, s->dec_lut the causal bucket table for [T, T].  tokens i64 [B]: decoder input at position t;
 * kv_cache bf16 [num_decoder_layers][B][T][2*H*d_kv] (caller-owned, k then v per position); logits f32 [B, vocab] out;
 * next_ids i64 [B] (optional) = argmax of the logits.  At t == 0 the cross-attention keys/values are projected from the
 * encoder output of the same step state (vlt5_encoder_fwd + prototype rows must have been written to the workspace). */
int vlt5_decoder_step(const vlt5_config* c, const vlt5_step* s, const long long* tokens, int t, void* kv_cache, float* logits,
                      long long* next_ids, void* stream);
/* The same step with HF generate's greedy-search loop body (vqa_model.py:112-116 -> GenerationMixin.greedy_search) on the device,
 * so that the host only enqueues steps: after the vocabulary projection the emitted token of row b is argmax(logits_b), or pad_id
 * once the row has produced eos_id; it is written to out_tokens[b*out_ld + t + 1], done[b] is updated, and the input embedding row
 * and relative-position bias row of step t + 1 are prepared in the workspace -- `tokens` is only read at t == 0, steps must be
 * issued in order t = 0, 1, ... on one workspace.  logits / next_ids may be NULL. */
typedef struct {
    const long long* tokens;       /* [B] decoder input at position 0 (decoder_start_token_id) */
    int t;
    void* kv_cache;                /* as for vlt5_decoder_step */
    float* logits;                 /* [B, vocab] out, or NULL */
    long long* next_ids;           /* [B] out: argmax of the logits (before the pad rule), or NULL */
    long long* out_tokens; long long out_ld;    /* [B, out_ld >= T]: column t + 1 receives the emitted token */
    int* done;                     /* [B] in/out: 1 once the row has emitted eos_id */
    int eos_id, pad_id;
    int* t_dev;                    /* optional device-side step index (one int, 0 before the first step): every launch of a step reads the cache
                                      slot / key count / output column from it and the step increments it -- `t` then only tells the first step
                                      (t == 0: encoder-side keys | values, first input row) from the others (any t != 0), whose launches do not
                                      depend on t: capture ONE such call in a HIP graph and replay it per token (vqa_model.greedy_generate) */
} vlt5_greedy_desc;
int vlt5_decoder_step_greedy(const vlt5_config* c, const vlt5_step* s, const vlt5_greedy_desc* g, void* stream);
/* 1 if this configuration / shape / tuning record decodes through the decode kernels (vlt5_decoder_step_greedy is then available;
 * vlt5_decoder_step uses them as well) */
int vlt5_decode_fast_supported(const vlt5_config* c, const vlt5_step* s);

/* ---- the kernels of a decoding step on their own (csrc/decode.hip) ---------------------------------------------------------
 * vlt5_decode_linear: out[m, n] = act(rowscale[m] * alpha * sum_k A[m, k] W[n, k] (+ resid[m, n])) for a FEW rows (a decoding batch)
 * against a bf16 weight [N, K] -- nn.Linear of T5Attention q/k/v/o, T5DenseReluDense wi/wo, lm_head on one token per sample.  A is
 * bf16 (x_bf16), or the f32 residual stream (x_f32) with the T5LayerNorm in front of the projection folded in: operand
 * bf16(x * norm_w), rows scaled by rsqrt(mean(x^2) + norm_eps).  Outputs: f32 (out_f32) and / or bf16 (out_bf16; columns >=
 * split_col are routed to out_bf16_2 -- the k | v columns of a fused q|k|v projection straight into a cache slot).  K % 32 == 0 and
 * K / 32 = KS * NW with KS in {1, 2, 6, 8 (, 12, 16 for bf16 A)}, NW in {1, 2, 4, 8} (vlt5_decode_linear_supported); N % 4 == 0.
 * argmax_val / argmax_idx (optional, [rows][vlt5_decode_linear_tiles(...)]): first maximum of every row inside every column tile.
 * A norm split between two launches (what the decode step chains): the launch that writes residual-stream rows (out_f32, N % 16 == 0)
 * also emits next_xn_bf16 = bf16(out * next_norm_w[n]) and next_ssq[m][N / 16] = the sum of out^2 of every 16-column fragment; the
 * projection behind that norm then takes x_bf16 = next_xn_bf16 with row_ssq = next_ssq, n_row_ssq = N / 16 (a multiple of 4, <= 64)
 * and scales its rows by rsqrt(sum_j row_ssq[m][j] / K + norm_eps) -- the same arithmetic as the folded form on half the bytes. */
typedef struct {
    const float* x_f32; const void* x_bf16; long long ldx;
    const float* norm_w; float norm_eps;
    const void* w_bf16; int rows, N, K;
    float alpha;                                   /* 0 = 1 */
    void* out_bf16; long long ld_out_bf16; int split_col; void* out_bf16_2; long long ld_out_bf16_2;
    float* out_f32; long long ld_out_f32;
    const float* resid; long long ld_resid;
    int relu;
    float* argmax_val; int* argmax_idx;
    const float* next_norm_w; void* next_xn_bf16; long long ld_next_xn; float* next_ssq;      /* producer side of a split norm */
    const float* row_ssq; int n_row_ssq;                                                       /* consumer side (with x_bf16) */
} vlt5_decode_linear_desc;
int vlt5_decode_linear(const vlt5_decode_linear_desc* d, void* stream);
int vlt5_decode_linear_supported(int K, int norm_folded);
int vlt5_decode_linear_tiles(int rows, int N, int K, int norm_folded);     /* norm_folded: 1 = x_f32 + norm_w, 0 = x_bf16, 2 = x_bf16 + row_ssq */
/* attention core for ONE query per (sample, head) (Tq == 1, no dropout, no causal mask: every cached key is visible): softmax in
 * f32 over <= 64 keys, d_kv in {16, 32, 64}; vlt5_attn_desc as for vlt5_attn_fwd with q_sb the sample stride of q, k_sb == v_sb,
 * k_st == v_st, bias [H][1][bias_k] (bias_q == 1), key_mask [B][Tk]; lse is not written. */
int vlt5_decode_attn(const vlt5_attn_desc* d, void* stream);
int vlt5_decoder_bwd(const vlt5_config* c, const vlt5_step* s, void* stream);
int vlt5_encoder_bwd(const vlt5_config* c, const vlt5_step* s, void* stream);

#ifdef __cplusplus
}
#endif
#endif
