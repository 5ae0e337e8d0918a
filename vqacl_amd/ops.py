"""Thin torch-tensor wrappers over the per-kernel C ABI (used by the parity tests and the host-side model code).

Every function enqueues on torch's current HIP stream.  Tensors must live on the GPU; nothing here computes
on the host and nothing falls back to torch ops.
"""
import ctypes as C

import torch

from . import _lib as L
from ._lib import AttnDesc, GemmDesc, check, lib, ptr, stream_ptr

BF16 = torch.bfloat16


def _need(t, dtype=None):
    if not t.is_cuda:
        raise L.Vlt5Error("tensor must be on the GPU")
    if dtype is not None and t.dtype != dtype:
        raise L.Vlt5Error(f"expected {dtype}, got {t.dtype}")
    return t


def gemm(A, B, M, N, K, *, a_kmajor=False, b_kmajor=False, out=None, out_f32=False, alpha=1.0, bias=None, relu=False,
         resid=None, gate=None, gate_scale=1.0, drop_p=0.0, drop_seed=0, accum=False, split_k=1, tile=(0, 0),
         lda=None, ldb=None, ldc=None, batch=1, batch_strides=(0, 0, 0), emit=None, norm=None, tuning=None, relu_bits_out=None,
         gate_bits=None):
    """C[M,N] = epi(alpha * sum_k A[m,k] B[n,k]); A,B bf16 2-D tensors, k-major flags as in vlt5_gemm_desc.
    tuning: a vlt5_tuning record (_lib.make_tuning) for the launch's policy switches.
    emit = (w_norm f32 [N], xw bf16 [M,N], partials f32 [M,32]): producer side of a folded T5 RMS norm -- returns (out, nparts);
    norm = (partials, nparts, d, eps, rstd_out or None): consumer side."""
    g, out, _keep = gemm_desc(A, B, M, N, K, a_kmajor=a_kmajor, b_kmajor=b_kmajor, out=out, out_f32=out_f32, alpha=alpha, bias=bias,
                              relu=relu, resid=resid, gate=gate, gate_scale=gate_scale, drop_p=drop_p, drop_seed=drop_seed,
                              accum=accum, split_k=split_k, tile=tile, lda=lda, ldb=ldb, ldc=ldc, batch=batch,
                              batch_strides=batch_strides)
    if emit is not None:
        g.emit_norm_w, g.emit_xw_bf16, g.emit_partials = ptr(emit[0]), ptr(emit[1]), ptr(emit[2])
    if norm is not None:
        g.norm_partials, g.norm_nparts, g.norm_d, g.norm_eps, g.norm_rstd_out = ptr(norm[0]), norm[1], norm[2], norm[3], ptr(norm[4])
    if tuning is not None:
        g.tuning = C.pointer(tuning)
    if relu_bits_out is not None:          # uint8 [M, >= N/8]: the ReLU sign bits of the stored values (vlt5_gemm_desc.relu_bits_out)
        g.relu_bits_out, g.ld_bits = ptr(relu_bits_out), relu_bits_out.stride(0)
    if gate_bits is not None:              # ... and the same bits as the gate of a hidden-gradient GEMM, instead of `gate`
        g.gate_bits, g.ld_bits = ptr(gate_bits), gate_bits.stride(0)
    check(lib().vlt5_gemm_bf16(C.byref(g), stream_ptr()), "vlt5_gemm_bf16")
    if emit is not None:
        return out, g.emit_nparts
    return out


def gemm_desc(A, B, M, N, K, *, a_kmajor=False, b_kmajor=False, out=None, out_f32=False, alpha=1.0, bias=None, relu=False,
              resid=None, gate=None, gate_scale=1.0, drop_p=0.0, drop_seed=0, accum=False, split_k=1, tile=(0, 0),
              lda=None, ldb=None, ldc=None, batch=1, batch_strides=(0, 0, 0)):
    """The filled vlt5_gemm_desc of `gemm` without launching it: (desc, out, keep-alive) -- for launch loops that must not pay the
    descriptor construction per call (bench.py)."""
    _need(A, BF16), _need(B, BF16)
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32 if out_f32 else BF16)
    out_f32 = out.dtype == torch.float32
    g = GemmDesc()
    g.A, g.B, g.C = ptr(A), ptr(B), ptr(out)
    g.M, g.N, g.K = M, N, K
    g.lda = lda if lda is not None else A.stride(0)
    g.ldb = ldb if ldb is not None else B.stride(0)
    g.ldc = ldc if ldc is not None else out.stride(0)
    g.a_kmajor, g.b_kmajor = int(a_kmajor), int(b_kmajor)
    g.alpha = alpha
    g.bias = ptr(bias)
    g.resid, g.ldr = ptr(resid), (resid.stride(0) if resid is not None else 0)
    g.gate, g.ldg, g.gate_scale = ptr(gate), (gate.stride(0) if gate is not None else 0), gate_scale
    g.drop_p, g.drop_seed = drop_p, drop_seed
    g.relu, g.out_f32, g.accum = int(relu), int(out_f32), int(accum)
    ws = None
    if split_k > 1:
        ws = torch.empty(lib().vlt5_gemm_workspace_bytes(M, g.ldc, split_k), device=A.device, dtype=torch.uint8)
        g.split_k, g.workspace = split_k, ptr(ws)
    g.tile_m, g.tile_n = tile
    g.batch = batch
    g.batch_stride_a, g.batch_stride_b, g.batch_stride_c = batch_strides
    return g, out, (A, B, ws, bias, resid, gate)


def layernorm_fwd(x, w, eps=1e-6, want_f32=False, drop_p=0.0, drop_seed=0):
    rows, d = x.shape
    yb = torch.empty(rows, d, device=x.device, dtype=BF16)
    yf = torch.empty(rows, d, device=x.device, dtype=torch.float32) if want_f32 else None
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    check(lib().vlt5_layernorm_fwd(ptr(_need(x, torch.float32)), ptr(w), ptr(yb), ptr(yf), ptr(rstd), rows, d, eps, drop_p,
                                   drop_seed, 0, 0, stream_ptr()), "vlt5_layernorm_fwd")
    return yb, yf, rstd


def layernorm_bwd(dy, x, w, rstd, dx=None, drop_p=0.0, drop_seed=0, want_bf16=False, dx_drop_p=0.0, dx_drop_seed=0,
                  deferred_reduce=False):
    rows, d = x.shape
    accum = dx is not None
    if dx is None:
        dx = torch.empty_like(x)
    dw = torch.empty(d, device=x.device, dtype=torch.float32)
    part = torch.empty(lib().vlt5_layernorm_bwd_blocks(rows), d, device=x.device, dtype=torch.float32)
    dxb = torch.empty(rows, d, device=x.device, dtype=BF16) if want_bf16 else None
    check(lib().vlt5_layernorm_bwd(ptr(dy), ptr(x), ptr(w), ptr(rstd), ptr(dx), ptr(None if deferred_reduce else dw), ptr(part),
                                   rows, d, int(accum), 0, drop_p, drop_seed, 0, 0, ptr(dxb), dx_drop_p, dx_drop_seed,
                                   stream_ptr()), "vlt5_layernorm_bwd")
    if deferred_reduce:                 # the engine's way: one multi-job reduction launch for many norms
        off = (L.c_ll * 1)(0)
        nb = (L.c_i * 1)(part.shape[0])
        check(lib().vlt5_colsum_multi(ptr(part), ptr(dw), off, nb, 1, part.shape[0], d, stream_ptr()), "vlt5_colsum_multi")
    if want_bf16:
        return dx, dw, dxb
    return dx, dw


def _attn_desc(q, k, v, H, dk, bias, key_mask, mask_value, causal, drop_p, drop_seed):
    B, Tq = q.shape[:2]
    Tk = k.shape[1]
    a = AttnDesc()
    a.q, a.k, a.v = ptr(_need(q, BF16)), ptr(_need(k, BF16)), ptr(_need(v, BF16))
    a.q_sb, a.q_st = q.stride(0), q.stride(1)
    a.k_sb, a.k_st = k.stride(0), k.stride(1)
    a.v_sb, a.v_st = v.stride(0), v.stride(1)
    if bias is not None:
        a.bias, a.bias_q, a.bias_k = ptr(bias), bias.shape[1], bias.shape[2]
    a.key_mask, a.mask_value, a.causal = ptr(key_mask), mask_value, int(causal)
    a.B, a.H, a.Tq, a.Tk, a.dk = B, H, Tq, Tk, dk
    a.drop_p, a.drop_seed = drop_p, drop_seed
    return a


def attn_fwd(q, k, v, H, dk, bias=None, key_mask=None, mask_value=-10000.0, causal=False, drop_p=0.0, drop_seed=0):
    """q [B,Tq,H*dk], k/v [B,Tk,H*dk] bf16 (any strides along batch/token) -> ctx [B,Tq,H*dk] bf16, lse [B,H,Tq]."""
    B, Tq = q.shape[:2]
    a = _attn_desc(q, k, v, H, dk, bias, key_mask, mask_value, causal, drop_p, drop_seed)
    ctx = torch.empty(B, Tq, H * dk, device=q.device, dtype=BF16)
    lse = torch.empty(B, H, Tq, device=q.device, dtype=torch.float32)
    a.ctx, a.o_sb, a.o_st, a.lse = ptr(ctx), ctx.stride(0), ctx.stride(1), ptr(lse)
    check(lib().vlt5_attn_fwd(C.byref(a), stream_ptr()), "vlt5_attn_fwd")
    return ctx, lse


def attn_bwd(q, k, v, d_ctx, lse, H, dk, bias=None, key_mask=None, mask_value=-10000.0, causal=False, drop_p=0.0,
             drop_seed=0, want_dbias=False):
    B, Tq = q.shape[:2]
    Tk = k.shape[1]
    a = _attn_desc(q, k, v, H, dk, bias, key_mask, mask_value, causal, drop_p, drop_seed)
    dq = torch.empty(B, Tq, H * dk, device=q.device, dtype=BF16)
    dk_ = torch.empty(B, Tk, H * dk, device=q.device, dtype=BF16)
    dv = torch.empty(B, Tk, H * dk, device=q.device, dtype=BF16)
    a.lse = ptr(lse)
    a.d_ctx, a.do_sb, a.do_st = ptr(_need(d_ctx, BF16)), d_ctx.stride(0), d_ctx.stride(1)
    a.dq, a.dq_sb, a.dq_st = ptr(dq), dq.stride(0), dq.stride(1)
    a.dk_, a.dk_sb, a.dk_st = ptr(dk_), dk_.stride(0), dk_.stride(1)
    a.dv, a.dv_sb, a.dv_st = ptr(dv), dv.stride(0), dv.stride(1)
    dbias = None
    if want_dbias and bias is not None:
        dbias = torch.zeros(B, H, bias.shape[1], bias.shape[2], device=q.device, dtype=torch.float32)
        a.dbias = ptr(dbias)
    check(lib().vlt5_attn_bwd(C.byref(a), stream_ptr()), "vlt5_attn_bwd")
    return dq, dk_, dv, dbias


def relbias_build(table, lut, H, Lq, Lk):
    bias = torch.empty(H, Lq, Lk, device=table.device, dtype=torch.float32)
    check(lib().vlt5_relbias_build(ptr(table), ptr(_need(lut, torch.int32)), ptr(bias), H, Lq, Lk, table.shape[0], stream_ptr()),
          "vlt5_relbias_build")
    return bias


def relbias_bwd(dS, lut, nbuckets):
    nmat, H, Lq, Lk = dS.shape
    dtable = torch.empty(nbuckets, H, device=dS.device, dtype=torch.float32)
    scratch = torch.empty(64 * H * Lq * Lk, device=dS.device, dtype=torch.float32)
    check(lib().vlt5_relbias_bwd(ptr(dS), ptr(lut), ptr(dtable), ptr(scratch), nmat, H, Lq, Lk, nbuckets, 0, stream_ptr()),
          "vlt5_relbias_bwd")
    return dtable


def cast_bf16(src, dst=None):
    if dst is None:
        dst = torch.empty(src.shape, device=src.device, dtype=BF16)
    check(lib().vlt5_cast_bf16(ptr(_need(src, torch.float32)), ptr(dst), src.numel(), stream_ptr()), "vlt5_cast_bf16")
    return dst


def ce_fwd(logits, labels):
    R, V = logits.shape
    loss = torch.empty(R, device=logits.device, dtype=torch.float32)
    lse = torch.empty(R, device=logits.device, dtype=torch.float32)
    check(lib().vlt5_ce_fwd(ptr(_need(logits, torch.float32)), ptr(_need(labels, torch.int64)), ptr(loss), ptr(lse), R, V,
                            stream_ptr()), "vlt5_ce_fwd")
    return loss, lse


def loss_reduce(loss_tok, labels, scores):
    B, T = labels.shape
    loss = torch.empty(1, device=labels.device, dtype=torch.float32)
    row_w = torch.empty(B * T, device=labels.device, dtype=torch.float32)
    check(lib().vlt5_loss_reduce(ptr(loss_tok), ptr(labels), ptr(scores), ptr(loss), ptr(row_w), B, T, stream_ptr()),
          "vlt5_loss_reduce")
    return loss, row_w


def ce_bwd(logits, labels, lse, row_w, gout=None):
    R, V = logits.shape
    d = torch.empty(R, V, device=logits.device, dtype=BF16)
    check(lib().vlt5_ce_bwd(ptr(logits), ptr(labels), ptr(lse), ptr(row_w), ptr(gout), ptr(d), R, V, stream_ptr()), "vlt5_ce_bwd")
    return d


def proto_pool(hidden, S, split):
    """hidden f32 [B, >=S, d] (batch stride may exceed S*d) -> poolQ, poolV [B,d]."""
    B, d = hidden.shape[0], hidden.shape[2]
    pq = torch.empty(B, d, device=hidden.device, dtype=torch.float32)
    pv = torch.empty(B, d, device=hidden.device, dtype=torch.float32)
    check(lib().vlt5_proto_pool(ptr(_need(hidden, torch.float32)), hidden.stride(0), B, S, d, split, ptr(pq), ptr(pv), stream_ptr()),
          "vlt5_proto_pool")
    return pq, pv


def proto_class_mean(pool, onehot):
    B, d = pool.shape
    Cn = onehot.shape[1]
    proto = torch.empty(Cn, d, device=pool.device, dtype=torch.float32)
    cnt = torch.empty(Cn, device=pool.device, dtype=torch.float32)
    onehot = _need(onehot.contiguous(), torch.float32)       # held in a local until the launch is enqueued
    check(lib().vlt5_proto_class_mean(ptr(pool), ptr(onehot), ptr(proto), ptr(cnt), B, Cn, d, stream_ptr()),
          "vlt5_proto_class_mean")
    return proto, cnt


def proto_retrieve(protos, pool, out_f32=None, sb=0, out_bf16=None, sb_bf16=0):
    B, d = pool.shape
    idx = torch.empty(B, device=pool.device, dtype=torch.int64)
    scratch = torch.empty(protos.shape[0], d, device=pool.device, dtype=torch.float32)
    check(lib().vlt5_proto_retrieve(ptr(_need(protos, torch.float32)), ptr(pool), ptr(idx), ptr(out_f32), sb, ptr(out_bf16), sb_bf16,
                                    ptr(scratch), B, protos.shape[0], d, stream_ptr()), "vlt5_proto_retrieve")
    return idx


def proto_memory_loss(pool, onehot, protos):
    out = torch.empty(1, device=pool.device, dtype=torch.float32)
    onehot = onehot.contiguous()
    check(lib().vlt5_proto_memory_loss(ptr(pool), ptr(onehot), ptr(protos), ptr(out), pool.shape[0], protos.shape[0],
                                       pool.shape[1], stream_ptr()), "vlt5_proto_memory_loss")
    return out


def drop_cast(src, drop_p=0.0, drop_seed=0):
    """bf16(dropout(src)) with the engine's counter-based mask (element index = row*cols + col)."""
    rows, cols = src.shape
    dst = torch.empty(rows, cols, device=src.device, dtype=BF16)
    check(lib().vlt5_drop_cast(ptr(_need(src, torch.float32)), ptr(dst), rows, cols, drop_p, drop_seed, stream_ptr()), "vlt5_drop_cast")
    return dst


def glu_fwd(u, ff, drop_p=0.0, drop_seed=0):
    """u bf16 [rows, 2*ff] -> h bf16 [rows, ff] = dropout(gelu_new(u[:, :ff]) * u[:, ff:])  (HF T5DenseGatedActDense)."""
    rows = u.shape[0]
    h = torch.empty(rows, ff, device=u.device, dtype=BF16)
    check(lib().vlt5_glu_fwd(ptr(_need(u, BF16)), ptr(h), rows, ff, drop_p, drop_seed, stream_ptr()), "vlt5_glu_fwd")
    return h


def glu_bwd(dh, u, ff, drop_p=0.0, drop_seed=0):
    rows = u.shape[0]
    du = torch.empty(rows, 2 * ff, device=u.device, dtype=BF16)
    check(lib().vlt5_glu_bwd(ptr(_need(dh, BF16)), ptr(_need(u, BF16)), ptr(du), rows, ff, drop_p, drop_seed, stream_ptr()), "vlt5_glu_bwd")
    return du


def qkv_attn_fwd(xn, wqkv, B, S, H, bias=None, key_mask=None, mask_value=-10000.0, drop_p=0.0, drop_seed=0):
    """Fused q|k|v projection + attention core (csrc/enc_attn.hip): xn bf16 [B*S, d], wqkv bf16 [3*H*64, d] ->
    qkv bf16 [B, S, 3*H*64], ctx bf16 [B, S, H*64], lse f32 [B, H, S]."""
    d = xn.shape[1]
    inner = H * 64
    qkv = torch.empty(B, S, 3 * inner, device=xn.device, dtype=BF16)
    a = _attn_desc(qkv[:, :, :inner], qkv[:, :, inner:2 * inner], qkv[:, :, 2 * inner:], H, 64, bias, key_mask, mask_value, False,
                   drop_p, drop_seed)
    ctx = torch.empty(B, S, inner, device=xn.device, dtype=BF16)
    lse = torch.empty(B, H, S, device=xn.device, dtype=torch.float32)
    a.ctx, a.o_sb, a.o_st, a.lse = ptr(ctx), ctx.stride(0), ctx.stride(1), ptr(lse)
    check(lib().vlt5_qkv_attn_fwd(ptr(_need(xn, BF16)), ptr(_need(wqkv, BF16)), ptr(qkv), C.byref(a), d, stream_ptr()), "vlt5_qkv_attn_fwd")
    return qkv, ctx, lse


def dec_attn_fused(xn, w, wo, B, T, H, k=None, v=None, bias=None, key_mask=None, mask_value=-10000.0, drop_p=0.0, drop_seed=0):
    """Fused decoder attention sublayer between the norms (csrc/dec_attn.hip).  Self-attention (k is None): w = [3*H*64, d] q|k|v rows,
    causal; cross-attention: w = [H*64, d] q rows, k / v = projected encoder-side keys / values [B, Tk, H*64] (any batch / token strides).
    xn bf16 [B*T, d], wo bf16 [d, H*64] -> (proj bf16 [B, T, 3*H*64] or [B, T, H*64], ctx bf16 [B, T, H*64], lse f32 [B, H, T],
    slabs f32 [H, B*T, d]: slab h = that head's share of the output projection)."""
    from ._lib import DecAttnDesc
    d = xn.shape[1]
    inner = H * 64
    dev = xn.device
    cross = k is not None
    proj = torch.empty(B, T, inner if cross else 3 * inner, device=dev, dtype=BF16)
    if cross:
        a = _attn_desc(proj, k, v, H, 64, bias, key_mask, mask_value, False, drop_p, drop_seed)
    else:
        a = _attn_desc(proj[:, :, :inner], proj[:, :, inner:2 * inner], proj[:, :, 2 * inner:], H, 64, bias, key_mask, mask_value, True,
                       drop_p, drop_seed)
    ctx = torch.empty(B, T, inner, device=dev, dtype=BF16)
    lse = torch.empty(B, H, T, device=dev, dtype=torch.float32)
    slabs = torch.empty(H, B * T, d, device=dev, dtype=torch.float32)
    a.ctx, a.o_sb, a.o_st, a.lse = ptr(ctx), ctx.stride(0), ctx.stride(1), ptr(lse)
    e = DecAttnDesc()
    e.xn_bf16, e.w_bf16, e.wo_bf16, e.proj_bf16 = ptr(_need(xn, BF16)), ptr(_need(w, BF16)), ptr(_need(wo, BF16)), ptr(proj)
    e.o_slabs, e.slab_stride, e.d_model, e.core = ptr(slabs), B * T * d, d, a
    fn = lib().vlt5_cross_attn_fwd if cross else lib().vlt5_dec_self_attn_fwd
    check(fn(C.byref(e), stream_ptr()), "vlt5_cross_attn_fwd" if cross else "vlt5_dec_self_attn_fwd")
    return proj, ctx, lse, slabs


def enc_attn_sublayer(x, ln_w, wqkv, wo, B, S, H, bias=None, key_mask=None, mask_value=-10000.0, eps=1e-6, drop_p=0.0, seeds=(0, 0)):
    """Forward of the whole encoder self-attention sublayer through `vlt5_enc_attn_fwd` (norm, fused q|k|v projection + core, output
    projection with dropout + residual).  x f32 [B*S, d]; wqkv bf16 [3*H*64, d]; wo bf16 [d, H*64].  Returns (x_out, saved) where
    `saved` feeds `enc_attn_sublayer_bwd`."""
    from ._lib import EncAttnDesc
    d = x.shape[1]
    inner = H * 64
    dev = x.device
    sv = dict(x=x, ln_w=ln_w, wqkv=wqkv, wo=wo, bias=bias, key_mask=key_mask,
              xn=torch.empty(B * S, d, device=dev, dtype=BF16), rstd=torch.empty(B * S, device=dev, dtype=torch.float32),
              qkv=torch.empty(B * S, 3 * inner, device=dev, dtype=BF16), ctx=torch.empty(B * S, inner, device=dev, dtype=BF16),
              lse=torch.empty(B, H, S, device=dev, dtype=torch.float32))
    out = torch.empty_like(x)
    e = EncAttnDesc()
    e.x, e.ln_w, e.wqkv_bf16, e.wo_bf16, e.x_out = ptr(_need(x, torch.float32)), ptr(ln_w), ptr(_need(wqkv, BF16)), ptr(_need(wo, BF16)), ptr(out)
    e.xn_bf16, e.rstd, e.qkv_bf16, e.ctx_bf16, e.lse = ptr(sv["xn"]), ptr(sv["rstd"]), ptr(sv["qkv"]), ptr(sv["ctx"]), ptr(sv["lse"])
    if bias is not None:
        e.bias, e.bias_q, e.bias_k = ptr(bias), bias.shape[1], bias.shape[2]
    e.key_mask, e.mask_value = ptr(key_mask), mask_value
    e.B, e.S, e.H, e.d_model, e.eps = B, S, H, d, eps
    e.drop_p, e.seed_probs, e.seed_out = drop_p, seeds[0], seeds[1]
    check(lib().vlt5_enc_attn_fwd(C.byref(e), stream_ptr()), "vlt5_enc_attn_fwd")
    sv["desc"] = e
    return out, sv


def enc_attn_sublayer_bwd(dy, sv, want_dscores=False):
    """Backward through `vlt5_enc_attn_bwd`: returns dx, d_wqkv, d_wo, d_ln_w (f32) and, optionally, the per-sample bias-block gradient."""
    from ._lib import EncAttnGrads
    e = sv["desc"]
    dev = dy.device
    g = EncAttnGrads()
    dx = torch.empty_like(dy)
    dwqkv = torch.empty(sv["wqkv"].shape, device=dev, dtype=torch.float32)
    dwo = torch.empty(sv["wo"].shape, device=dev, dtype=torch.float32)
    dln = torch.empty(sv["ln_w"].shape, device=dev, dtype=torch.float32)
    ds = None
    if want_dscores and sv["bias"] is not None:
        ds = torch.zeros(e.B, e.H, e.bias_q, e.bias_k, device=dev, dtype=torch.float32)
    ws = torch.empty(lib().vlt5_enc_attn_bwd_workspace_bytes(e.B, e.S, e.H, e.d_model), device=dev, dtype=torch.uint8)
    g.dy, g.dx, g.d_wqkv, g.d_wo, g.d_ln_w, g.d_scores = ptr(_need(dy, torch.float32)), ptr(dx), ptr(dwqkv), ptr(dwo), ptr(dln), ptr(ds)
    check(lib().vlt5_enc_attn_bwd(C.byref(e), C.byref(g), ptr(ws), stream_ptr()), "vlt5_enc_attn_bwd")
    return dx, dwqkv, dwo, dln, ds
