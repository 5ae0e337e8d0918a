"""ctypes binding of libvlt5_hip.so (include/vlt5_hip.h).

The library is the product's only compute path: if it is missing or a call fails, this module raises --
there is no eager/CPU fallback anywhere in `vqacl_amd`.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: it loads the HIP runtime the library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VLT5_LIB") or os.path.join(_HERE, "libvlt5_hip.so")   # VLT5_LIB: A/B runs of kernel variants

c_f = C.c_float
c_i = C.c_int
c_ll = C.c_longlong
c_u32 = C.c_uint32
vp = C.c_void_p


class Tuning(C.Structure):
    """vlt5_tuning: experiment switches, every field 0 = the library's default (include/vlt5_hip.h)."""
    _fields_ = [("fold_norm", c_i), ("fold_norm_dec", c_i), ("fused_attn", c_i), ("fused_heads", c_i), ("dec_fused", c_i),
                ("enc_cut", c_i), ("wgrad_shadow", c_i), ("wgrad_grouped", c_i), ("gemm_t128_kmkm", c_i), ("gemm_t256_km", c_i),
                ("gemm_t256_min", c_i), ("gemm_dec_tall", c_i), ("gemm_split_kmin", c_i), ("decode_fast", c_i), ("decode_split_norm", c_i),
                ("gemm_rmkm_tile", c_i), ("gemm_rmrm_f32_tile", c_i), ("gemm_split_cap", c_i), ("decode_nfrag", c_i), ("ffn_gate_bits", c_i)]


# environment variable -> (field, how its value maps): the library itself reads no environment; the host reads these ONCE per model
# (tuning_from_env) and hands the record to every engine call (vlt5_step.tuning).  on/off switches: "0" / "1" -> 1 (off) / 2 (on).
_TUNING_ENV = {"VLT5_FOLD_NORM": ("fold_norm", "onoff"), "VLT5_FOLD_NORM_DEC": ("fold_norm_dec", "onoff"),
               "VLT5_FUSED_ATTN": ("fused_attn", "onoff"), "VLT5_FUSED_HEADS": ("fused_heads", "int"), "VLT5_DEC_FUSED": ("dec_fused", "onoff"),
               "VLT5_ENC_CUT": ("enc_cut", "int"), "VLT5_WGRAD_SHADOW": ("wgrad_shadow", "onoff"), "VLT5_WGRAD_GROUPED": ("wgrad_grouped", "onoff"),
               "VLT5_GEMM_T128_KMKM": ("gemm_t128_kmkm", "int"), "VLT5_GEMM_T256_KM": ("gemm_t256_km", "int"),
               "VLT5_GEMM_T256_MIN": ("gemm_t256_min", "int"), "VLT5_GEMM_DEC_TALL": ("gemm_dec_tall", "onoff"),
               "VLT5_GEMM_SPLIT_KMIN": ("gemm_split_kmin", "int"), "VLT5_DECODE_FAST": ("decode_fast", "onoff"),
               "VLT5_DECODE_SPLIT_NORM": ("decode_split_norm", "onoff"), "VLT5_GEMM_RMKM_TILE": ("gemm_rmkm_tile", "int"),
               "VLT5_GEMM_RMRM_F32_TILE": ("gemm_rmrm_f32_tile", "int"), "VLT5_GEMM_SPLIT_CAP": ("gemm_split_cap", "int"), "VLT5_DECODE_NFRAG": ("decode_nfrag", "int"),
               "VLT5_FFN_GATE_BITS": ("ffn_gate_bits", "onoff")}


def make_tuning(**fields):
    """A vlt5_tuning record: keyword = field name; on/off fields take True / False (-> 2 / 1), the others integers."""
    t = Tuning()
    kinds = {f: k for f, k in _TUNING_ENV.values()}
    for name, v in fields.items():
        if name not in kinds:
            raise ValueError(f"unknown tuning field {name!r}")
        setattr(t, name, (2 if v else 1) if kinds[name] == "onoff" and isinstance(v, bool) else int(v))     # (ints pass through: a third setting)
    return t


def tuning_from_env(environ=None):
    """The experiment switches the environment asks for (A/B scripts under tools/): read here, once, never inside the library."""
    environ = os.environ if environ is None else environ
    t = Tuning()
    for var, (field, kind) in _TUNING_ENV.items():
        if var in environ:
            v = int(environ[var])
            setattr(t, field, ((2 if v else 1) if v in (0, 1) else v) if kind == "onoff" else v)     # on/off: "0" / "1"; a larger value is a third setting, as is
    return t


class GemmDesc(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("M", c_i), ("N", c_i), ("K", c_i), ("lda", c_i), ("ldb", c_i), ("ldc", c_i),
                ("a_kmajor", c_i), ("b_kmajor", c_i), ("alpha", c_f), ("bias", vp), ("resid", vp), ("ldr", c_i),
                ("gate", vp), ("ldg", c_i), ("gate_scale", c_f), ("drop_p", c_f), ("drop_seed", c_u32),
                ("relu", c_i), ("out_f32", c_i), ("accum", c_i), ("split_k", c_i), ("workspace", vp),
                ("tile_m", c_i), ("tile_n", c_i), ("batch", c_i), ("batch_stride_a", c_ll), ("batch_stride_b", c_ll),
                ("batch_stride_c", c_ll), ("defer_reduce", c_i), ("split_used", c_i), ("c_bf16_copy", vp),
                ("grouped_with", vp), ("emit_norm_w", vp), ("emit_xw_bf16", vp), ("emit_partials", vp), ("emit_nparts", c_i),
                ("norm_partials", vp), ("norm_nparts", c_i), ("norm_d", c_i), ("norm_eps", c_f), ("norm_rstd_out", vp),
                ("sumsq", vp), ("sumsq_batch_stride", c_ll), ("tuning", C.POINTER(Tuning)),
                ("relu_bits_out", vp), ("gate_bits", vp), ("ld_bits", c_i)]


class AttnDesc(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("q_sb", c_ll), ("q_st", c_ll), ("k_sb", c_ll), ("k_st", c_ll),
                ("v_sb", c_ll), ("v_st", c_ll), ("ctx", vp), ("o_sb", c_ll), ("o_st", c_ll), ("lse", vp),
                ("bias", vp), ("bias_q", c_i), ("bias_k", c_i), ("key_mask", vp), ("mask_value", c_f), ("causal", c_i),
                ("B", c_i), ("H", c_i), ("Tq", c_i), ("Tk", c_i), ("dk", c_i), ("drop_p", c_f), ("drop_seed", c_u32),
                ("d_ctx", vp), ("do_sb", c_ll), ("do_st", c_ll), ("dq", vp), ("dk_", vp), ("dv", vp),
                ("dq_sb", c_ll), ("dq_st", c_ll), ("dk_sb", c_ll), ("dk_st", c_ll), ("dv_sb", c_ll), ("dv_st", c_ll),
                ("dbias", vp), ("fused_heads", c_i)]


class DecAttnDesc(C.Structure):
    _fields_ = [("xn_bf16", vp), ("w_bf16", vp), ("wo_bf16", vp), ("proj_bf16", vp), ("o_slabs", vp), ("slab_stride", c_ll),
                ("d_model", c_i), ("core", AttnDesc)]


class EncAttnDesc(C.Structure):
    _fields_ = [("x", vp), ("ln_w", vp), ("wqkv_bf16", vp), ("wo_bf16", vp), ("x_out", vp), ("xn_bf16", vp), ("rstd", vp),
                ("qkv_bf16", vp), ("ctx_bf16", vp), ("lse", vp), ("bias", vp), ("bias_q", c_i), ("bias_k", c_i), ("key_mask", vp),
                ("mask_value", c_f), ("B", c_i), ("S", c_i), ("H", c_i), ("d_model", c_i), ("eps", c_f), ("drop_p", c_f),
                ("seed_probs", c_u32), ("seed_out", c_u32)]


class EncAttnGrads(C.Structure):
    _fields_ = [("dy", vp), ("dx", vp), ("d_wqkv", vp), ("d_wo", vp), ("d_ln_w", vp), ("d_scores", vp)]


class StackInputsDesc(C.Structure):
    _fields_ = [("mask_ids", vp), ("mask", vp), ("B", c_i), ("L", c_i), ("S", c_i), ("rel_table", vp), ("lut", vp), ("bias", vp),
                ("H", c_i), ("Lq", c_i), ("Lk", c_i), ("ids", vp), ("labels", vp), ("ids_out", vp), ("T", c_i), ("start_id", c_i),
                ("pad_id", c_i), ("table", vp), ("out", vp), ("out_sb", c_ll), ("out_st", c_ll), ("d", c_i), ("vocab", c_i),
                ("drop_p", c_f), ("drop_seed", c_u32), ("drop_rows", c_i), ("drop_row0", c_i)]


class ProtoHeadDesc(C.Structure):
    _fields_ = [("hidden", vp), ("hidden_sb", c_ll), ("B", c_i), ("S", c_i), ("d", c_i), ("split", c_i), ("poolQ", vp), ("poolV", vp),
                ("onehotQ", vp), ("onehotV", vp), ("Qproto", vp), ("Vproto", vp), ("Qnum", vp), ("Vnum", vp), ("qmem", vp),
                ("qmem_initialised", c_i), ("first", c_i), ("task", c_i), ("update", c_i), ("alpha", c_f), ("beta", c_f),
                ("CQ", c_i), ("CV", c_i), ("idxQ", vp), ("idxV", vp), ("out_f32", vp), ("out_sb", c_ll), ("out_bf16", vp),
                ("out_sb_bf16", c_ll), ("scratch", vp), ("packed", vp), ("phase", c_i)]


class GemmTimingRec(C.Structure):
    _fields_ = [("M", c_i), ("N", c_i), ("K", c_i), ("batch", c_i), ("tile_m", c_i), ("tile_n", c_i), ("a_kmajor", c_i),
                ("b_kmajor", c_i), ("splits", c_i), ("workgroups", c_i), ("out_f32", c_i), ("ms", c_f), ("M2", c_i), ("N2", c_i), ("K2", c_i), ("batch2", c_i)]


class Config(C.Structure):
    _fields_ = [("d_model", c_i), ("d_kv", c_i), ("num_heads", c_i), ("d_ff", c_i), ("num_layers", c_i),
                ("num_decoder_layers", c_i), ("vocab", c_i), ("rel_buckets", c_i), ("feat_dim", c_i), ("n_images", c_i),
                ("pad_id", c_i), ("dec_start_id", c_i), ("n_ques", c_i), ("n_cate", c_i), ("eps", c_f), ("dropout", c_f),
                ("gated_act", c_i)]


class Step(C.Structure):
    _fields_ = [("B", c_i), ("L", c_i), ("V", c_i), ("T", c_i), ("training", c_i), ("seed", c_u32),
                ("params", vp), ("params_bf16", vp), ("grads", vp), ("workspace", vp), ("workspace_bytes", c_ll),
                ("vis_feats", vp), ("boxes", vp), ("input_ids", vp), ("labels", vp), ("scores", vp),
                ("enc_lut", vp), ("dec_lut", vp), ("gout", vp), ("d_loss_tok", vp), ("events", C.POINTER(vp)), ("n_events", c_i),
                ("wait_events", C.POINTER(vp)), ("n_wait_events", c_i),
                ("feat_store", vp), ("box_store", vp), ("feat_slots", vp), ("n_slots", c_ll),
                ("side_stream", vp), ("side_events", C.POINTER(vp)), ("n_side_events", c_i), ("grads_bf16", vp),
                ("gnorm_partials", vp), ("defer_decoder_wgrads", c_i), ("tuning", C.POINTER(Tuning)), ("release_plan_id", c_i)]


class DecAttnGrads(C.Structure):
    _fields_ = [("d_out", vp), ("d_xn", vp), ("d_w", vp), ("d_wo", vp), ("d_proj", vp), ("dk", vp), ("dv", vp), ("dkv_sb", c_ll),
                ("dkv_st", c_ll), ("d_scores", vp)]


class FfnDesc(C.Structure):
    _fields_ = [("x", vp), ("ln_w", vp), ("wi_bf16", vp), ("wo_bf16", vp), ("x_out", vp), ("xn_bf16", vp), ("rstd", vp), ("h_bf16", vp),
                ("u_bf16", vp), ("M", c_i), ("d_model", c_i), ("d_ff", c_i), ("gated", c_i), ("eps", c_f), ("drop_p", c_f),
                ("seed_hidden", c_u32), ("seed_out", c_u32), ("tuning", C.POINTER(Tuning))]


class FfnGrads(C.Structure):
    _fields_ = [("dy", vp), ("dx", vp), ("d_wi", vp), ("d_wo", vp), ("d_ln_w", vp)]


class LmheadCeDesc(C.Structure):
    _fields_ = [("x_bf16", vp), ("emb_bf16", vp), ("labels", vp), ("logits", vp), ("loss_tok", vp), ("lse", vp), ("rows", c_i),
                ("d_model", c_i), ("vocab", c_i), ("tuning", C.POINTER(Tuning))]


class LmheadCeGrads(C.Structure):
    _fields_ = [("d_loss_tok", vp), ("dlogits_bf16", vp), ("d_x", vp), ("d_emb", vp), ("accum_d_emb", c_i)]


class GreedyDesc(C.Structure):
    _fields_ = [("tokens", vp), ("t", c_i), ("kv_cache", vp), ("logits", vp), ("next_ids", vp), ("out_tokens", vp), ("out_ld", c_ll),
                ("done", vp), ("eos_id", c_i), ("pad_id", c_i), ("t_dev", vp)]


class DecodeLinearDesc(C.Structure):
    _fields_ = [("x_f32", vp), ("x_bf16", vp), ("ldx", c_ll), ("norm_w", vp), ("norm_eps", c_f), ("w_bf16", vp), ("rows", c_i), ("N", c_i),
                ("K", c_i), ("alpha", c_f), ("out_bf16", vp), ("ld_out_bf16", c_ll), ("split_col", c_i), ("out_bf16_2", vp),
                ("ld_out_bf16_2", c_ll), ("out_f32", vp), ("ld_out_f32", c_ll), ("resid", vp), ("ld_resid", c_ll), ("relu", c_i),
                ("argmax_val", vp), ("argmax_idx", vp), ("next_norm_w", vp), ("next_xn_bf16", vp), ("ld_next_xn", c_ll), ("next_ssq", vp),
                ("row_ssq", vp), ("n_row_ssq", c_i)]


# name -> (restype, argtypes); every symbol declared in include/vlt5_hip.h
PROTOTYPES = {
    "vlt5_abi_version": (c_i, []),
    "vlt5_build_flags": (c_i, []),
    "vlt5_grad_release_plan": (c_i, [C.POINTER(Config), C.POINTER(Tuning), c_i, C.POINTER(c_i), c_i, C.POINTER(c_i)]),
    "vlt5_gemm_bf16": (c_i, [C.POINTER(GemmDesc), vp]),
    "vlt5_gemm_workspace_bytes": (c_ll, [c_i, c_i, c_i]),
    "vlt5_gemm_auto_split": (c_i, [c_i, c_i, c_i, c_ll]),
    "vlt5_gemm_auto_split_tuned": (c_i, [c_i, c_i, c_i, c_ll, C.POINTER(Tuning)]),
    "vlt5_gemm_timing_enable": (c_i, [c_i]),
    "vlt5_gemm_timing_collect": (c_i, [C.POINTER(GemmTimingRec), c_i]),
    "vlt5_layernorm_fwd": (c_i, [vp, vp, vp, vp, vp, c_i, c_i, c_f, c_f, c_u32, c_i, c_i, vp]),
    "vlt5_layernorm_fwd_slabs": (c_i, [vp, c_i, c_ll, vp, vp, c_f, c_u32, vp, vp, vp, vp, c_i, c_i, c_f, c_f, c_u32, c_i, c_i, vp]),
    "vlt5_layernorm_bwd": (c_i, [vp, vp, vp, vp, vp, vp, vp, c_i, c_i, c_i, c_i, c_f, c_u32, c_i, c_i, vp, c_f, c_u32, vp]),
    "vlt5_layernorm_bwd_slabs": (c_i, [vp, c_i, c_ll, vp, vp, vp, vp, vp, vp, c_i, c_i, c_i, c_i, c_f, c_u32, c_i, c_i, vp, c_f, c_u32, vp]),
    "vlt5_layernorm_bwd_full": (c_i, [vp, c_i, c_ll, vp, vp, vp, vp, vp, vp, c_i, c_i, c_i, c_i, c_f, c_u32, c_i, c_i, vp, c_f, c_u32, vp, vp]),
    "vlt5_encoder_late_layers": (c_i, [c_i]),
    "vlt5_encoder_late_layers_tuned": (c_i, [c_i, C.POINTER(Tuning)]),
    "vlt5_decoder_buckets_late": (c_i, [C.POINTER(Config), C.POINTER(Tuning), c_i]),
    "vlt5_side_stream_create": (c_i, [C.POINTER(vp)]),
    "vlt5_side_stream_destroy": (c_i, [vp]),
    "vlt5_feat_store_put": (c_i, [vp, vp, vp, c_i, vp, vp, c_ll, c_i, c_i, vp]),
    "vlt5_feat_gather": (c_i, [vp, vp, vp, c_ll, vp, vp, c_i, c_i, c_i, vp]),
    "vlt5_colsum_multi": (c_i, [vp, vp, C.POINTER(c_ll), C.POINTER(c_i), c_i, c_i, c_i, vp]),
    "vlt5_layernorm_bwd_blocks": (c_i, [c_i]),
    "vlt5_attn_fwd": (c_i, [C.POINTER(AttnDesc), vp]),
    "vlt5_qkv_attn_fwd": (c_i, [vp, vp, vp, C.POINTER(AttnDesc), c_i, vp]),
    "vlt5_qkv_attn_fwd_norm": (c_i, [vp, vp, vp, C.POINTER(AttnDesc), c_i, vp, c_i, c_f, vp, vp]),
    "vlt5_dec_self_attn_fwd": (c_i, [C.POINTER(DecAttnDesc), vp]),
    "vlt5_cross_attn_fwd": (c_i, [C.POINTER(DecAttnDesc), vp]),
    "vlt5_dec_attn_fused_ok": (c_i, [c_i, c_i, c_i, c_i]),
    "vlt5_dec_attn_bwd_workspace_bytes": (c_ll, [c_i, c_i, c_i, c_i, c_i]),
    "vlt5_dec_self_attn_bwd": (c_i, [C.POINTER(DecAttnDesc), C.POINTER(DecAttnGrads), vp, vp]),
    "vlt5_cross_attn_bwd": (c_i, [C.POINTER(DecAttnDesc), C.POINTER(DecAttnGrads), vp, vp]),
    "vlt5_ffn_fwd": (c_i, [C.POINTER(FfnDesc), vp]),
    "vlt5_ffn_bwd_workspace_bytes": (c_ll, [c_i, c_i, c_i, c_i]),
    "vlt5_ffn_bwd": (c_i, [C.POINTER(FfnDesc), C.POINTER(FfnGrads), vp, vp]),
    "vlt5_lmhead_ce_fwd": (c_i, [C.POINTER(LmheadCeDesc), vp]),
    "vlt5_lmhead_ce_bwd": (c_i, [C.POINTER(LmheadCeDesc), C.POINTER(LmheadCeGrads), vp]),
    "vlt5_enc_attn_fwd": (c_i, [C.POINTER(EncAttnDesc), vp]),
    "vlt5_enc_attn_bwd_workspace_bytes": (c_ll, [c_i, c_i, c_i, c_i]),
    "vlt5_enc_attn_bwd": (c_i, [C.POINTER(EncAttnDesc), C.POINTER(EncAttnGrads), vp, vp]),
    "vlt5_attn_bwd": (c_i, [C.POINTER(AttnDesc), vp]),
    "vlt5_relbias_build": (c_i, [vp, vp, vp, c_i, c_i, c_i, c_i, vp]),
    "vlt5_relbias_bwd": (c_i, [vp, vp, vp, vp, c_i, c_i, c_i, c_i, c_i, c_i, vp]),
    "vlt5_embed_fwd": (c_i, [vp, vp, vp, c_ll, c_ll, c_i, c_i, c_i, c_i, c_f, c_u32, c_i, c_i, vp]),
    "vlt5_embed_bwd_scratch_bytes": (c_ll, [c_i, c_i, c_i]),
    "vlt5_embed_bwd": (c_i, [vp, vp, c_ll, c_ll, vp, c_i, c_i, c_i, c_i, c_f, c_u32, c_i, c_i, vp, vp]),
    "vlt5_mirror_rows_bf16": (c_i, [vp, vp, c_i, c_i, vp, c_i, vp, c_i, c_i, vp]),
    "vlt5_shift_right": (c_i, [vp, vp, c_i, c_i, c_i, c_i, vp]),
    "vlt5_stack_inputs_fwd": (c_i, [C.POINTER(StackInputsDesc), vp]),
    "vlt5_build_mask": (c_i, [vp, vp, c_i, c_i, c_i, c_i, vp]),
    "vlt5_vis_embed_fwd": (c_i, [vp] * 9 + [c_ll, c_ll, vp, vp, c_i, c_i, c_i, c_i, c_f, c_f, c_u32, c_i, c_i, vp]),
    "vlt5_vis_embed_bwd": (c_i, [vp, c_ll, c_ll] + [vp] * 11 + [c_i, c_i, c_i, c_i, c_f, c_u32, c_i, c_i, vp]),
    "vlt5_vis_embed_bwd_blocks": (c_i, [c_i]),
    "vlt5_vis_grad_scatter": (c_i, [vp, vp, vp, vp, vp, vp, vp, c_i, c_i, vp]),
    "vlt5_colsum": (c_i, [vp, vp, c_i, c_i, c_i, c_i, vp]),
    "vlt5_ce_fwd": (c_i, [vp, vp, vp, vp, c_i, c_i, vp]),
    "vlt5_loss_reduce": (c_i, [vp, vp, vp, vp, vp, c_i, c_i, vp]),
    "vlt5_argmax_rows": (c_i, [vp, c_i, c_i, vp, vp]),
    "vlt5_ce_bwd": (c_i, [vp, vp, vp, vp, vp, vp, c_i, c_i, vp]),
    "vlt5_proto_pool": (c_i, [vp, c_ll, c_i, c_i, c_i, c_i, vp, vp, vp]),
    "vlt5_proto_class_mean": (c_i, [vp, vp, vp, vp, c_i, c_i, c_i, vp]),
    "vlt5_proto_stats_pack": (c_i, [vp, vp, vp, vp, vp, c_i, c_i, c_i, c_i, vp]),
    "vlt5_proto_update": (c_i, [vp] * 9 + [c_i, c_i, c_i, c_f, c_f, c_i, c_i, c_i, vp]),
    "vlt5_proto_retrieve": (c_i, [vp, vp, vp, vp, c_ll, vp, c_ll, vp, c_i, c_i, c_i, vp]),
    "vlt5_proto_memory_loss": (c_i, [vp, vp, vp, vp, c_i, c_i, c_i, vp]),
    "vlt5_proto_head_fwd": (c_i, [C.POINTER(ProtoHeadDesc), vp]),
    "vlt5_sqnorm": (c_i, [vp, c_ll, vp, vp, c_i, vp]),
    "vlt5_gnorm_finish": (c_i, [vp, c_ll, vp, C.POINTER(c_ll), C.POINTER(c_ll), c_i, vp, vp, vp]),
    "vlt5_gnorm_slots": (c_ll, [C.POINTER(Config)]),
    "vlt5_sqnorm_blocks": (c_i, [c_ll]),
    "vlt5_adamw_step": (c_i, [vp, vp, vp, vp, vp, c_ll, c_f, c_f, c_f, c_f, c_f, c_i, vp, c_f, c_i, vp]),
    "vlt5_sqnorm_g16": (c_i, [vp, c_f, c_ll, vp, vp, c_i, vp]),
    "vlt5_adamw_step_g16": (c_i, [vp, vp, c_f, vp, vp, vp, c_ll, c_f, c_f, c_f, c_f, c_f, c_i, vp, c_f, c_i, vp]),
    "vlt5_cast_bf16": (c_i, [vp, vp, c_ll, vp]),
    "vlt5_cast_f32": (c_i, [vp, vp, c_ll, c_f, vp]),
    "vlt5_scale_add": (c_i, [vp, vp, c_f, c_f, c_ll, vp]),
    "vlt5_drop_cast": (c_i, [vp, vp, c_ll, c_i, c_f, c_u32, vp]),
    "vlt5_glu_fwd": (c_i, [vp, vp, c_ll, c_i, c_f, c_u32, vp]),
    "vlt5_glu_bwd": (c_i, [vp, vp, vp, c_ll, c_i, c_f, c_u32, vp]),
    "vlt5_layout_count": (c_i, [C.POINTER(Config)]),
    "vlt5_layout_get": (c_i, [C.POINTER(Config), c_i, C.c_char_p, c_i, C.POINTER(c_ll), C.POINTER(c_i), C.POINTER(c_i),
                              C.POINTER(c_i), C.POINTER(c_i), C.POINTER(c_i)]),
    "vlt5_layout_total": (c_ll, [C.POINTER(Config)]),
    "vlt5_layout_buckets": (c_i, [C.POINTER(Config)]),
    "vlt5_workspace_bytes": (c_ll, [C.POINTER(Config), c_i, c_i, c_i, c_i]),
    "vlt5_workspace_offset": (c_ll, [C.POINTER(Config), c_i, c_i, c_i, c_i, c_i]),
    "vlt5_encoder_fwd": (c_i, [C.POINTER(Config), C.POINTER(Step), vp]),
    "vlt5_decoder_fwd": (c_i, [C.POINTER(Config), C.POINTER(Step), vp]),
    "vlt5_decoder_step": (c_i, [C.POINTER(Config), C.POINTER(Step), vp, c_i, vp, vp, vp, vp]),
    "vlt5_decoder_step_greedy": (c_i, [C.POINTER(Config), C.POINTER(Step), C.POINTER(GreedyDesc), vp]),
    "vlt5_decode_fast_supported": (c_i, [C.POINTER(Config), C.POINTER(Step)]),
    "vlt5_decode_linear": (c_i, [C.POINTER(DecodeLinearDesc), vp]),
    "vlt5_decode_linear_supported": (c_i, [c_i, c_i]),
    "vlt5_decode_linear_tiles": (c_i, [c_i, c_i, c_i, c_i]),
    "vlt5_decode_attn": (c_i, [C.POINTER(AttnDesc), vp]),
    "vlt5_decoder_bwd": (c_i, [C.POINTER(Config), C.POINTER(Step), vp]),
    "vlt5_encoder_bwd": (c_i, [C.POINTER(Config), C.POINTER(Step), vp]),
}

WS_ENC_OUT, WS_ENC_EXT, WS_LOGITS, WS_LOSS_TOK, WS_LOSS, WS_ENC_MASK_EXT, WS_DEC_OUT = range(7)

_lib = None
ABI_VERSION = 8
BUILD_FLAG_NAMES = {1: "ENC_DGRAD_HOT_A (input-gradient GEMM reads one row: wrong gradients)", 2: "ATTN_BWD_NO_STORE (attention backward "
                    "stores nothing: wrong gradients)", 4: "TIMELINE (phase stamps written into user buffers)"}


class Vlt5Error(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Vlt5Error(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no fallback path)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)          # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.vlt5_abi_version() != ABI_VERSION:
            raise Vlt5Error(f"{LIB_PATH}: ABI version {L.vlt5_abi_version()}, this package needs {ABI_VERSION} -- rebuild it "
                            "(python -c 'import __graft_entry__ as g; g.build()')")
        flags = L.vlt5_build_flags()
        if flags and os.environ.get("VLT5_ALLOW_EXPERIMENT") != "1":
            # (tools/build_variant.sh builds: upper-bound measurements whose RESULTS ARE WRONG, timeline stamps into user buffers)
            raise Vlt5Error(f"{LIB_PATH} is an EXPERIMENT build (vlt5_build_flags() = {flags}: "
                            + ", ".join(n for b, n in BUILD_FLAG_NAMES.items() if flags & b)
                            + "): its results are not the product's.  Set VLT5_ALLOW_EXPERIMENT=1 to load it for a measurement.")
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        kind = {1001: "bad argument", 1002: "alignment (contiguous dims must be multiples of 8)",
                1003: "the caller's gradient release plan (vlt5_step.release_plan_id) is not the order this call completes the buckets in "
                      "-- tuning / side stream changed after the data-parallel wrapper was built?"}.get(rc, f"hipError {rc}")
        raise Vlt5Error(f"{what} failed: {kind}")


def stream_ptr():
    return vp(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return vp(0)
    return vp(t.data_ptr())
