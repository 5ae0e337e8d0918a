"""GPU parity tests of the greedy-decoding kernels (csrc/decode.hip), through the C ABI.

Checker: plain fp32 torch math on the same seeded inputs (bf16 operands rounded BEFORE the reference computation), the tiled
GEMM / attention launches of the training path (vlt5_tuning.decode_fast = off), and the CPU oracle's greedy loop under the top-2
margin rule for the integer outputs (tokens).  Reference path being replaced: HF generate -> VLT5.forward(decoder_input_ids[:, -1:],
past_key_values), VL-T5/src/vqa_model.py:68-121, src/modeling_t5_our.py:608-629, 715-772.
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    from vqacl_amd import _lib
    _lib.lib()
    return torch.device("cuda")


def rel_max_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


@pytest.mark.parametrize("rows,N,K,mode", [
    (80, 2304, 768, "norm_split"),       # norm -> q | k | v of t5-base, k | v routed into a cache slot
    (80, 768, 768, "bf16_resid"),        # attention output projection + residual
    (80, 3072, 768, "norm_relu"),        # norm -> wi + ReLU
    (80, 768, 3072, "bf16_resid"),       # wo + residual (reduction over 8 waves)
    (80, 32200, 768, "norm_argmax"),     # final norm -> rescale -> tied lm_head, per-tile first maximum
    (4, 32200, 768, "norm_argmax"),      # BASELINE configs[0] batch: one ragged row block, narrow column tiles
    (4, 2304, 768, "norm_split"),
    (4, 768, 768, "bf16_resid"),
    (4, 768, 768, "norm_relu"),
    (4, 3072, 768, "norm_relu"),
    (4, 768, 3072, "bf16_resid"),
    (80, 768, 768, "norm_relu"),         # norm -> cross-attention q (one fragment per column tile)
    (80, 1024, 4096, "bf16_resid"),      # t5-large wo
    (32, 4096, 1024, "norm_relu"),       # t5-large wi
    (100, 2304, 768, "norm_split"),      # the reference's --valid_batch_size: seven row blocks (one ragged) -> the narrow-tile geometry of round 5
    (100, 768, 768, "bf16_resid"),       #   (two fragments per tile: 24 x 7 workgroups = one round of the chip)
    (100, 3072, 768, "norm_relu"),       #   (one fragment per tile)
    (100, 768, 3072, "bf16_resid"),
    (100, 32200, 768, "norm_argmax"),    #   (the vocabulary projection keeps its resident form up to eight row blocks)
    (160, 768, 768, "norm_relu"),        # ten row blocks
    (320, 3072, 768, "norm_relu"),
    (160, 32200, 768, "norm_argmax"),    # more than eight row blocks: the row-walking form of the vocabulary projection
    (5, 192, 64, "norm_split"),          # tiny configuration, ragged row block
    (33, 64, 128, "bf16_resid"),
    (17, 400, 64, "norm_argmax"),
])
def test_decode_linear_vs_f32_reference(dev, rows, N, K, mode):
    from vqacl_amd import _lib as L
    from vqacl_amd._lib import lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(rows * 131 + N * 7 + K)
    norm = mode.startswith("norm")
    assert lib().vlt5_decode_linear_supported(K, int(norm)) == 1
    W = (torch.randn(N, K, generator=g) * K ** -0.5 + torch.arange(N)[:, None] * 1e-5).to(BF)      # asymmetric: a transposed write cannot pass
    d = L.DecodeLinearDesc()
    keep = []
    if norm:
        x = torch.randn(rows, K, generator=g) * 3.0
        w = 1.0 + 0.2 * torch.randn(K, generator=g)
        xw = (x * w).to(BF).float()
        rstd = torch.rsqrt((x * x).mean(dim=1, keepdim=True) + 1e-6)
        alpha = K ** -0.5 if "argmax" in mode else 1.0
        ref = (xw @ W.float().t()) * rstd * alpha
        xd, wd = x.to(dev), w.to(dev)
        d.x_f32, d.norm_w, d.norm_eps, d.alpha = ptr(xd), ptr(wd), 1e-6, alpha
        keep += [xd, wd]
    else:
        xb = torch.randn(rows, K, generator=g).to(BF)
        ref = xb.float() @ W.float().t()
        xd = xb.to(dev)
        d.x_bf16 = ptr(xd)
        keep += [xd]
    d.ldx, d.rows, d.N, d.K = K, rows, N, K
    Wd = W.to(dev)
    d.w_bf16 = ptr(Wd)
    out_f = torch.full((rows, N), float("nan"), device=dev)
    d.out_f32, d.ld_out_f32 = ptr(out_f), N
    if "resid" in mode:
        r = torch.randn(rows, N, generator=g)
        rd = r.to(dev)
        d.resid, d.ld_resid = ptr(rd), N
        ref = ref + r
        keep.append(rd)
    if "relu" in mode:
        d.relu = 1
        ref = ref.clamp(min=0)
    out_b = out_b2 = None
    if "split" in mode:                   # columns [0, N/3) -> a [rows, N/3] buffer, the rest -> slot 3 of a [rows][8][2N/3] cache
        sc = N // 3
        out_b = torch.zeros(rows, sc, device=dev, dtype=BF)
        cache = torch.zeros(rows, 8, N - sc, device=dev, dtype=BF)
        d.out_bf16, d.ld_out_bf16, d.split_col = ptr(out_b), sc, sc
        d.out_bf16_2, d.ld_out_bf16_2 = C.c_void_p(cache.data_ptr() + 3 * (N - sc) * 2), 8 * (N - sc)
        out_b2 = cache
    elif "relu" in mode:
        out_b = torch.zeros(rows, N, device=dev, dtype=BF)
        d.out_bf16, d.ld_out_bf16 = ptr(out_b), N
    tiles = 0
    if "argmax" in mode:
        tiles = lib().vlt5_decode_linear_tiles(rows, N, K, 1)
        assert tiles > 0
        pv = torch.full((rows, tiles), float("nan"), device=dev)
        pi = torch.full((rows, tiles), -1, device=dev, dtype=torch.int32)
        d.argmax_val, d.argmax_idx = ptr(pv), ptr(pi)
    assert lib().vlt5_decode_linear(C.byref(d), stream_ptr()) == 0
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    err = float((out_f.cpu() - ref).abs().max())
    assert err <= 2e-3 * scale + 1e-4, f"f32 output: max err {err:.4g} against max |ref| {scale:.4g}"      # accumulation order only
    if out_b is not None:
        nb = out_b.shape[1]
        assert float((out_b.float().cpu() - ref[:, :nb]).abs().max()) <= 1e-2 * scale, "bf16 output"
        # the bf16 output is the f32 output rounded (same registers)
        assert torch.equal(out_b.cpu(), out_f[:, :nb].to(BF).cpu())
    if out_b2 is not None:
        sc = N // 3
        assert torch.equal(out_b2[:, 3].cpu(), out_f[:, sc:].to(BF).cpu()), "columns behind split_col land in the cache slot"
        assert float(out_b2[:, :3].abs().sum()) == 0 and float(out_b2[:, 4:].abs().sum()) == 0, "no other slot is touched"
    if tiles:
        # integer output: the first maximum of every row, bit-exact against torch.argmax over the logits the kernel itself wrote
        best = pv.max(dim=1, keepdim=True).values
        cand = torch.where(pv == best, pi, torch.full_like(pi, 2 ** 30))
        assert torch.equal(cand.min(dim=1).values.long(), out_f.argmax(dim=1))
        width = next(16 * f for f in (1, 2, 3, 4) if (N + 16 * f - 1) // (16 * f) == tiles)        # columns per tile
        assert bool(((pi >= 0) & (pi < N)).all()) and bool((pi.long() // width == torch.arange(tiles, device=dev)[None, :]).all())


@pytest.mark.parametrize("rows,d_model,K_in,N_out,mode", [
    (80, 768, 768, 2304, "plain"),       # attention output projection + residual -> [norm] -> q | k | v (t5-base)
    (80, 768, 3072, 3072, "relu"),       # wo + residual -> [norm] -> wi + ReLU of the next sublayer
    (80, 768, 3072, 32200, "argmax"),    # last wo + residual -> [final norm] -> rescale -> lm_head with the per-tile argmax (row-walking kernel)
    (4, 768, 768, 32200, "argmax"),      # BASELINE configs[0] batch
    (33, 1024, 1024, 1024, "plain"),     # t5-large width: 64 partial sums per row
    (100, 768, 768, 32200, "argmax"),    # seven row blocks: seven waves in the resident vocabulary kernel
    (140, 768, 768, 32200, "argmax"),    # nine row blocks: the row-walking kernel on the bf16 operand, two chunks of row blocks
    (20, 1024, 1024, 32128, "argmax"),   # t5-large width and vocabulary: the resident kernel with 16 k-tiles per row
    (5, 64, 64, 192, "plain"),           # tiny configuration
])
def test_decode_linear_norm_split_between_two_launches(dev, rows, d_model, K_in, N_out, mode):
    """The decode step's chain: the launch that writes a residual-stream row (x_bf16 W^T + resid) also emits bf16(row * w_norm) and the
    row's partial sums of squares; the projection behind the norm consumes them (x_bf16 + row_ssq).  Checked against an f32 reference
    and against the folded form (x_f32 + norm_w) of the same projection on the row the producer wrote."""
    from vqacl_amd import _lib as L
    from vqacl_amd._lib import lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(rows * 31 + d_model + K_in * 3 + N_out)
    xb = torch.randn(rows, K_in, generator=g).to(BF)
    W1 = (torch.randn(d_model, K_in, generator=g) * K_in ** -0.5).to(BF)
    r = torch.randn(rows, d_model, generator=g) * 2.0
    wn = 1.0 + 0.2 * torch.randn(d_model, generator=g)
    W2 = (torch.randn(N_out, d_model, generator=g) * d_model ** -0.5 + torch.arange(N_out)[:, None] * 1e-6).to(BF)
    xbd, W1d, rd, wnd, W2d = xb.to(dev), W1.to(dev), r.to(dev), wn.to(dev), W2.to(dev)
    parts = d_model // 16
    y = torch.full((rows, d_model), float("nan"), device=dev)
    xn = torch.zeros(rows, d_model, device=dev, dtype=BF)
    ssq = torch.full((rows, parts), float("nan"), device=dev)
    p = L.DecodeLinearDesc()
    p.x_bf16, p.ldx, p.w_bf16, p.rows, p.N, p.K = ptr(xbd), K_in, ptr(W1d), rows, d_model, K_in
    p.out_f32, p.ld_out_f32, p.resid, p.ld_resid = ptr(y), d_model, ptr(rd), d_model
    p.next_norm_w, p.next_xn_bf16, p.ld_next_xn, p.next_ssq = ptr(wnd), ptr(xn), d_model, ptr(ssq)
    assert lib().vlt5_decode_linear(C.byref(p), stream_ptr()) == 0
    torch.cuda.synchronize()
    y_ref = xb.float() @ W1.float().t() + r
    assert float((y.cpu() - y_ref).abs().max()) <= 2e-3 * float(y_ref.abs().max()) + 1e-4
    # what the producer emits is a function of the row it wrote: bit-exact operand, sums of squares to f32 rounding
    assert torch.equal(xn.cpu(), (y.cpu() * wn).to(BF))
    assert torch.allclose(ssq.sum(dim=1).cpu(), (y.cpu() ** 2).sum(dim=1), rtol=1e-5)
    assert torch.allclose(ssq.cpu(), (y.cpu() ** 2).view(rows, parts, 16).sum(dim=2), rtol=1e-5, atol=1e-6)

    def consumer(split):
        c = L.DecodeLinearDesc()
        c.w_bf16, c.rows, c.N, c.K, c.ldx, c.norm_eps = ptr(W2d), rows, N_out, d_model, d_model, 1e-6
        if split:
            c.x_bf16, c.row_ssq, c.n_row_ssq = ptr(xn), ptr(ssq), parts
        else:
            c.x_f32, c.norm_w = ptr(y), ptr(wnd)
        out = torch.full((rows, N_out), float("nan"), device=dev)
        c.out_f32, c.ld_out_f32 = ptr(out), N_out
        extra = {}
        if mode == "relu":
            c.relu = 1
            ob = torch.zeros(rows, N_out, device=dev, dtype=BF)
            c.out_bf16, c.ld_out_bf16 = ptr(ob), N_out
            extra["ob"] = ob
        if mode == "argmax":
            c.alpha = d_model ** -0.5
            tiles = lib().vlt5_decode_linear_tiles(rows, N_out, d_model, 2 if split else 1)
            assert tiles > 0
            pv = torch.full((rows, tiles), float("nan"), device=dev)
            pi = torch.full((rows, tiles), -1, device=dev, dtype=torch.int32)
            c.argmax_val, c.argmax_idx = ptr(pv), ptr(pi)
            extra.update(pv=pv, pi=pi)
        assert lib().vlt5_decode_linear(C.byref(c), stream_ptr()) == 0
        torch.cuda.synchronize()
        return out, extra

    out_s, ex_s = consumer(True)
    out_f, ex_f = consumer(False)
    yc = y.cpu()
    ref = ((yc * wn).to(BF).float() @ W2.float().t()) * torch.rsqrt((yc * yc).mean(dim=1, keepdim=True) + 1e-6)
    if mode == "argmax":
        ref = ref * d_model ** -0.5
    if mode == "relu":
        ref = ref.clamp(min=0)
    scale = float(ref.abs().max())
    assert float((out_s.cpu() - ref).abs().max()) <= 2e-3 * scale + 1e-4
    # same operand bits, same MFMA order; only the order of the sum of squares differs (rstd to an ulp or two)
    assert float((out_s - out_f).abs().max()) <= 1e-5 * scale + 1e-6
    if mode == "relu":
        assert torch.equal(ex_s["ob"].cpu(), out_s.to(BF).cpu())
    if mode == "argmax":
        pv, pi = ex_s["pv"], ex_s["pi"]
        best = pv.max(dim=1, keepdim=True).values
        cand = torch.where(pv == best, pi, torch.full_like(pi, 2 ** 30))
        assert torch.equal(cand.min(dim=1).values.long(), out_s.argmax(dim=1))


def test_decode_linear_rejects_bad_arguments(dev):
    from vqacl_amd import _lib as L
    from vqacl_amd._lib import lib, ptr, stream_ptr
    x = torch.zeros(8, 96, device=dev, dtype=BF)
    W = torch.zeros(64, 96, device=dev, dtype=BF)
    o = torch.zeros(8, 64, device=dev)
    d = L.DecodeLinearDesc()
    d.x_bf16, d.ldx, d.w_bf16, d.rows, d.N, d.K, d.out_f32, d.ld_out_f32 = ptr(x), 96, ptr(W), 8, 64, 96, ptr(o), 64
    assert lib().vlt5_decode_linear_supported(96, 0) == 0 and lib().vlt5_decode_linear(C.byref(d), stream_ptr()) in (1001, 1002)   # K % 64 != 0
    assert lib().vlt5_decode_linear_supported(80, 0) == 0
    assert all(lib().vlt5_decode_linear_supported(K, f) == 1 for K in (64, 128, 512, 768, 1024, 2048, 3072, 4096) for f in (0, 1))
    d.K, d.ldx = 64, 64
    d.out_f32 = None
    assert lib().vlt5_decode_linear(C.byref(d), stream_ptr()) == 1001                                                  # no output
    assert lib().vlt5_decode_linear(None, stream_ptr()) == 1001
    # split norm: the producer needs the f32 output, the norm weights and the partials buffer; the consumer a bf16 operand and 4 | n <= 64
    d.out_f32 = ptr(o)
    xn = torch.zeros(8, 64, device=dev, dtype=BF)
    d.next_xn_bf16, d.ld_next_xn = ptr(xn), 64
    assert lib().vlt5_decode_linear(C.byref(d), stream_ptr()) == 1001                                                  # no norm weights / partials
    d.next_xn_bf16 = None
    ss = torch.zeros(8, 6, device=dev)
    d.row_ssq, d.n_row_ssq = ptr(ss), 6
    assert lib().vlt5_decode_linear(C.byref(d), stream_ptr()) == 1001                                                  # 6 parts: not a multiple of 4
    xf = torch.zeros(8, 64, device=dev)
    wf = torch.ones(64, device=dev)
    d.x_bf16, d.x_f32, d.norm_w, d.n_row_ssq = None, ptr(xf), ptr(wf), 4
    assert lib().vlt5_decode_linear(C.byref(d), stream_ptr()) == 1001                                                  # partials with an f32 operand


@pytest.mark.parametrize("B,H,dk,Tk,kind", [(80, 12, 64, 1, "self"), (80, 12, 64, 7, "self"), (80, 12, 64, 20, "self"), (80, 12, 64, 64, "self"),
                                            (80, 12, 64, 58, "cross"), (5, 16, 64, 41, "cross"), (3, 4, 16, 58, "cross"), (3, 4, 16, 6, "self"),
                                            (7, 8, 32, 33, "cross"), (7, 8, 32, 17, "self")])
def test_decode_attention_core_vs_f32_reference(dev, B, H, dk, Tk, kind):
    """One query per (sample, head): softmax(q K^T + bias + mask) V in f32 over the cached keys (HF T5Attention.forward with a
    past_key_value: no 1/sqrt(d) scaling), strided k | v as they lie in the self-attention cache / the stacked cross-K/V buffer."""
    from vqacl_amd import _lib as L
    from vqacl_amd._lib import lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(B * 17 + H * 5 + dk + Tk)
    inner, Tcap = H * dk, 64
    q = (torch.randn(B, inner, generator=g) * 0.5).to(BF)
    kv = (torch.randn(B, Tcap, 2 * inner, generator=g) * 0.7).to(BF)            # k | v per position, as the cache holds them
    a = L.AttnDesc()
    qd, kvd = q.to(dev), kv.to(dev)
    ctx = torch.zeros(B, inner, device=dev, dtype=BF)
    a.q, a.k, a.v = ptr(qd), ptr(kvd), C.c_void_p(kvd.data_ptr() + inner * 2)
    a.q_sb, a.k_sb, a.k_st, a.v_sb, a.v_st = inner, Tcap * 2 * inner, 2 * inner, Tcap * 2 * inner, 2 * inner
    a.ctx, a.o_sb = ptr(ctx), inner
    a.B, a.H, a.Tq, a.Tk, a.dk = B, H, 1, Tk, dk
    K = kv[:, :Tk, :inner].float().view(B, Tk, H, dk)
    V = kv[:, :Tk, inner:].float().view(B, Tk, H, dk)
    s = torch.einsum("bhd,bthd->bht", q.float().view(B, H, dk), K)
    keep = []
    if kind == "self":
        bias = torch.randn(H, 1, Tcap, generator=g)
        bd = bias.to(dev)
        a.bias, a.bias_q, a.bias_k = ptr(bd), 1, Tcap
        s = s + bias[None, :, 0, :Tk]
        keep.append(bd)
    else:
        mask = (torch.rand(B, Tk, generator=g) > 0.2).float()
        mask[:, 0] = 1.0
        md = mask.to(dev)
        a.key_mask, a.mask_value = ptr(md), -1e9
        s = s + (1.0 - mask)[:, None, :] * -1e9
        keep.append(md)
    ref = torch.einsum("bht,bthd->bhd", torch.softmax(s, dim=-1), V).reshape(B, inner)
    assert lib().vlt5_decode_attn(C.byref(a), stream_ptr()) == 0
    torch.cuda.synchronize()
    err = float((ctx.float().cpu() - ref).abs().max())
    assert err <= 1e-2 * float(ref.abs().max()) + 1e-3, f"context: max err {err:.4g}"          # bf16 output rounding + f32 summation order
    # the stand-alone core of the training path on the same inputs (P rounded to bf16 there, f32 here)
    a2 = L.AttnDesc()
    C.memmove(C.byref(a2), C.byref(a), C.sizeof(a))
    ctx2 = torch.zeros(B, inner, device=dev, dtype=BF)
    a2.ctx = ptr(ctx2)
    a2.q_st = a2.o_st = inner
    if lib().vlt5_attn_fwd(C.byref(a2), stream_ptr()) == 0:        # (head widths the tiled core does not serve are only checked against f32)
        torch.cuda.synchronize()
        assert float((ctx2.float() - ctx.float()).abs().max()) <= 2e-2 * float(ref.abs().max()) + 1e-3
    # argument checks
    a.Tq = 2
    assert lib().vlt5_decode_attn(C.byref(a), stream_ptr()) == 1001
    a.Tq, a.Tk = 1, 65
    assert lib().vlt5_decode_attn(C.byref(a), stream_ptr()) == 1001


def _base_model(dev, seed, B, L=20, T=5, boost=1.0):
    from oracle import ref_cpu as R
    from test_gpu_model import make_model
    ocfg = R.Cfg(dropout=0.0)
    params = R.init_params(ocfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    for k in params:
        if params[k].dim() == 1:
            params[k] = params[k] + 0.1 * torch.randn(params[k].shape, generator=g)
        elif boost != 1.0 and k.startswith("decoder.") and (k.endswith(".o.weight") or k.endswith(".wo.weight")):
            # a freshly initialised tied-embedding model just echoes its input token (the embedding in the residual stream dominates the
            # logits): stronger sublayer outputs make the decoded tokens vary, so that the integer checks below have something to bite on
            params[k] = params[k] * boost
    batch = R.synthetic_batch(ocfg, B=B, L=L, V=36, T=T, seed=seed + 2, task_id=0)
    return R, ocfg, params, batch, make_model(ocfg, params, dev)


def _step_logits(model, batch, dec_in, dev, fast):
    """T incremental steps through vlt5_decoder_step on the given decoder inputs: logits [T][B, vocab] and argmax ids."""
    from vqacl_amd import _lib as L
    from vqacl_amd import ops
    from vqacl_amd._lib import check, lib, ptr, stream_ptr
    model.tuning.decode_fast = 2 if fast else 1
    B, T = dec_in.shape
    feats, boxes = batch["vis_feats"].to(dev), batch["boxes"].to(dev)
    ids = batch["input_ids"].to(dev).contiguous()
    Lt, V = ids.shape[1], feats.shape[1]
    dims = (B, Lt, V, T)
    model._workspace(*dims)
    model.sync_bf16()
    st = dict(dims=dims, training=False, seed=0, feats=feats.float().contiguous(), boxes=boxes.float().contiguous(), input_ids=ids,
              labels=torch.zeros(B, T, dtype=torch.long, device=dev), enc_lut=model._lut(Lt, Lt, True), dec_lut=model._lut(T, T, False))
    c = model.cfg.c_struct()
    cs = model._make_step(st)
    assert lib().vlt5_decode_fast_supported(C.byref(c), C.byref(cs)) == int(fast)
    check(lib().vlt5_encoder_fwd(C.byref(c), C.byref(cs), stream_ptr()))
    S, Sx, d = Lt + V, Lt + V + 2, model.cfg.d_model
    enc_f32 = model._ws_view(c, dims, L.WS_ENC_OUT, torch.float32, (B, Sx, d))
    enc_b16 = model._ws_view(c, dims, L.WS_ENC_EXT, torch.bfloat16, (B, Sx, d))
    pq, pv = ops.proto_pool(enc_f32, S, model.L)
    model.proto.retrieve(pq, pv, enc_f32, enc_b16, S)
    inner = model.cfg.num_heads * model.cfg.d_kv
    cache = torch.zeros(model.cfg.num_decoder_layers, B, T, 2 * inner, device=dev, dtype=BF)
    logits = torch.empty(B, model.cfg.vocab_size, device=dev)
    nxt = torch.empty(B, dtype=torch.long, device=dev)
    outs, ids_out = [], []
    for t in range(T):
        tok = dec_in[:, t].contiguous().to(dev)
        check(lib().vlt5_decoder_step(C.byref(c), C.byref(cs), ptr(tok), t, ptr(cache), ptr(logits), ptr(nxt), stream_ptr()))
        outs.append(logits.clone())
        ids_out.append(nxt.clone())
    return outs, ids_out, cache


@pytest.mark.parametrize("B", [4, 80, 100])
def test_decode_kernels_against_the_tiled_path_at_base_size(dev, B):
    """VL-T5-base, the decode kernels against the tiled GEMM / attention launches of the training path on the same decoder inputs, step
    by step: logits within the stated bf16 tolerance of each other, the cache contents equal up to bf16 rounding, next-token ids equal
    wherever the top-2 logit margin exceeds twice the tolerance, and bit-exact against torch.argmax of the logits each path wrote."""
    R, ocfg, params, batch, model = _base_model(dev, 2024 + B, B)
    model.eval()
    T = 6
    g = torch.Generator().manual_seed(5)
    dec_in = torch.randint(2, 32000, (B, T), generator=g)
    dec_in[:, 0] = 0
    with torch.no_grad():
        tiled, ids_t, cache_t = _step_logits(model, batch, dec_in, dev, fast=False)
        fast, ids_f, cache_f = _step_logits(model, batch, dec_in, dev, fast=True)
    worst, checked = 0.0, 0
    for t in range(T):
        e = rel_max_err(fast[t], tiled[t])
        worst = max(worst, e)
        assert e < 1e-2, (t, e)
        assert torch.equal(ids_f[t], fast[t].argmax(dim=-1)) and torch.equal(ids_t[t], tiled[t].argmax(dim=-1))
        top = tiled[t].topk(2, dim=-1).values
        ok = (top[:, 0] - top[:, 1]) / tiled[t].abs().max(dim=-1).values > 2e-2
        assert torch.equal(ids_f[t][ok], ids_t[t][ok])
        checked += int(ok.sum())
    assert rel_max_err(cache_f, cache_t) < 2e-2
    # the norms folded into the projections (vlt5_tuning.decode_split_norm off) against the default, the norm split between two launches:
    # per projection the same operand bits and a sum of squares in another order (1e-5, kernel test above); through 12 layers the odd
    # flipped bf16 rounding of an activation grows to ~2e-3 of the logit range
    model.tuning.decode_split_norm = 1
    with torch.no_grad():
        folded, ids_n, cache_n = _step_logits(model, batch, dec_in, dev, fast=True)
    model.tuning.decode_split_norm = 0
    for t in range(T):
        assert rel_max_err(folded[t], fast[t]) < 5e-3, t
    assert rel_max_err(cache_n, cache_f) < 1e-2
    from test_gpu_model import parity_log
    parity_log(f"decode kernels vs tiled path (base, B={B}, {T} steps): logits rel max err {worst:.4g}, {checked} of {B * T} margin-gated "
               f"next-token ids equal")
    from test_gpu_model import check_pin
    check_pin(f"decode kernels vs tiled path B={B}/logits", worst, "logits")
    assert checked > 0


def test_greedy_generate_base_model_vs_oracle_and_bookkeeping(dev):
    """test_step / greedy_generate through the decode kernels at VL-T5-base size against the oracle's greedy loop (tokens exact under the
    top-2 margin rule), against the tiled path, and HF's bookkeeping on the device: a row that emitted EOS keeps emitting pad."""
    from test_gpu_model import check_greedy_tokens, oracle_greedy, parity_log
    R, ocfg, params, batch, model = _base_model(dev, 77, 4, L=12, boost=8.0)
    model.train()
    model.train_step(batch, 0, 0.5, 0.3)          # populate the prototypes
    model.eval()
    fb = (batch["vis_feats"], batch["boxes"])
    steps = 7
    st = R.PrototypeState(Q_prototype=model.Q_prototype.cpu().clone(), V_prototype=model.V_prototype.cpu().clone())
    torch.set_num_threads(8)
    ref_tok, margins = oracle_greedy(R, dict(params), st, ocfg, batch, steps)
    model.tuning.decode_fast = 2
    a = model.greedy_generate(batch["input_ids"], fb, max_length=steps + 1, eos_token_id=-1)
    model.tuning.decode_fast = 1
    b = model.greedy_generate(batch["input_ids"], fb, max_length=steps + 1, eos_token_id=-1)
    model.tuning.decode_fast = 0
    assert a.shape == b.shape == (4, steps + 1)
    ca, cut_a = check_greedy_tokens(a, ref_tok, margins, 4e-2, eos=-1, what="decode kernels vs oracle")
    cb, cut_b = check_greedy_tokens(b, ref_tok, margins, 4e-2, eos=-1, what="tiled path vs oracle")
    parity_log(f"greedy decode (base, B=4, {steps} steps): decode kernels {ca} / tiled path {cb} tokens bit-exact under the margin rule "
               f"({cut_a} / {cut_b} rows left the band)")
    assert ca >= 4
    # bookkeeping: take a token some row emits early as EOS -> that row is pad from the next step on, the other rows are unaffected until
    # they emit it themselves; identical to the host-side loop of the tiled path (torch.where / done flags)
    cands = [(r, t) for t in (2, 1, 3, 4, 5) for r in range(4) if int(a[r, t]) != model.cfg.pad_token_id]
    assert cands, f"no non-pad token emitted: {a.tolist()}"
    r0, t0 = cands[0]
    eos = int(a[r0, t0])
    model.tuning.decode_fast = 2
    fa = model.greedy_generate(batch["input_ids"], fb, max_length=steps + 1, eos_token_id=eos)
    model.tuning.decode_fast = 1
    fb_ = model.greedy_generate(batch["input_ids"], fb, max_length=steps + 1, eos_token_id=eos)
    model.tuning.decode_fast = 0
    assert int(fa[r0, t0]) == eos or eos in fa[r0, 1:t0].tolist()
    n = min(fa.shape[1], fb_.shape[1])
    same = (a[:, :n] == b[:, :n]).all(dim=1)                     # rows on which the two paths agree without EOS agree with it too
    assert torch.equal(fa[same][:, :n], fb_[same][:, :n])
    for r in range(4):                                           # every row: tokens up to its first EOS equal the EOS-free run, pad afterwards
        row, free = fa[r].tolist(), a[r].tolist()
        cut = row.index(eos) if eos in row[1:] else len(row) - 1
        assert row[:cut + 1] == free[:cut + 1] and all(x == model.cfg.pad_token_id for x in row[cut + 1:])


def test_greedy_generate_at_the_evaluation_batch_size_vs_oracle(dev):
    """The decode kernels at the benched evaluation shape (B = 80, L = 20) against the ORACLE's greedy loop -- not only against the tiled
    HIP path: every token whose top-2 logit margin exceeds twice the logits tolerance must be the oracle's (decode weights un-boosted
    rows included: the assertion is on the count of gated tokens, most of the 80 x 5)."""
    from test_gpu_model import check_greedy_tokens, oracle_greedy, parity_log
    R, ocfg, params, batch, model = _base_model(dev, 79, 80, L=20, boost=8.0)
    model.train()
    model.train_step(batch, 0, 0.5, 0.3)          # populate the prototypes
    model.eval()
    fb = (batch["vis_feats"], batch["boxes"])
    steps = 5
    st = R.PrototypeState(Q_prototype=model.Q_prototype.cpu().clone(), V_prototype=model.V_prototype.cpu().clone())
    torch.set_num_threads(16)
    ref_tok, margins = oracle_greedy(R, dict(params), st, ocfg, batch, steps)
    model.tuning.decode_fast = 2
    a = model.greedy_generate(batch["input_ids"], fb, max_length=steps + 1, eos_token_id=-1)
    model.tuning.decode_fast = 0
    assert a.shape == (80, steps + 1)
    ca, cut_a = check_greedy_tokens(a, ref_tok, margins, 4e-2, eos=-1, what="decode kernels vs oracle, B=80")
    parity_log(f"greedy decode (base, B=80, {steps} steps): decode kernels {ca} of {80 * steps} tokens bit-exact under the margin rule "
               f"({cut_a} rows left the band)")
    assert ca >= 200, (ca, cut_a)


def test_greedy_loop_replayed_from_a_graph_equals_the_enqueued_loop(dev):
    """greedy_generate replays the token-step from a HIP graph (device-side step index): the same tokens as enqueueing the launches every
    step, on repeated calls (the graph and its buffers are reused), with another batch in between, with an EOS that stops rows early, and
    after the parameters moved (the graph is keyed by what its raw pointers come from)."""
    R, ocfg, params, batch, model = _base_model(dev, 4242, 6, L=11, boost=8.0)
    model.eval()
    fb = (batch["vis_feats"], batch["boxes"])

    def run(graph, eos=-1, b=batch, f=fb, n=9):
        model.decode_graph = graph
        return model.greedy_generate(b["input_ids"], f, max_length=n, eos_token_id=eos).clone()

    ref = run(False)
    a1, a2 = run(True), run(True)
    assert torch.equal(ref, a1) and torch.equal(ref, a2)
    states = model._decode_states
    assert len(states) == 1 and next(iter(states.values()))["graph"] is not None
    # another batch of the same shape through the same graph
    batch2 = R.synthetic_batch(ocfg, B=6, L=11, V=36, T=5, seed=99)
    fb2 = (batch2["vis_feats"], batch2["boxes"])
    assert torch.equal(run(False, b=batch2, f=fb2), run(True, b=batch2, f=fb2))
    assert torch.equal(run(True), ref), "the first batch again"
    # an EOS some row emits early: pad afterwards, the loop may stop before max_length
    cands = [int(v) for v in ref[:, 2].tolist() if v != model.cfg.pad_token_id]
    if cands:
        assert torch.equal(run(False, eos=cands[0]), run(True, eos=cands[0]))
    # a different length is a different graph; the old one is still valid
    assert torch.equal(run(False, n=5), run(True, n=5)) and len(model._decode_states) >= 2
    # the parameters move: same shape key, the graph is re-captured
    with torch.no_grad():
        model._flat.mul_(1.0)
        p = dict(model.named_parameters())["decoder.block.0.layer.2.DenseReluDense.wo.weight"]
        p.add_(0.01 * torch.randn_like(p))
    model.sync_bf16()
    assert torch.equal(run(False), run(True))
    model.decode_graph = True


def test_failed_graph_capture_falls_back_to_enqueued_launches(dev, monkeypatch):
    """A capture that fails (e.g. a pin-memory thread of the evaluator's loader touching the runtime inside the capture window) must not
    abort evaluation: that shape decodes with enqueued launches -- the same kernels, so the same tokens -- and is not re-captured on every
    call.  The capture itself is requested in thread-local error mode."""
    import warnings
    R, ocfg, params, batch, model = _base_model(dev, 4243, 4, L=11, boost=8.0)
    model.eval()
    fb = (batch["vis_feats"], batch["boxes"])
    model.decode_graph = False
    ref = model.greedy_generate(batch["input_ids"], fb, max_length=8, eos_token_id=-1).clone()
    model.decode_graph = True
    seen = {}
    real_graph = torch.cuda.graph

    class Broken:
        def __init__(self, gr, *a, **kw):
            seen["mode"] = kw.get("capture_error_mode")
            seen["calls"] = seen.get("calls", 0) + 1

        def __enter__(self):
            raise RuntimeError("HIP error: operation failed due to a previous error during capture")

        def __exit__(self, *a):
            return False
    monkeypatch.setattr(torch.cuda, "graph", Broken)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        a = model.greedy_generate(batch["input_ids"], fb, max_length=8, eos_token_id=-1).clone()
        b = model.greedy_generate(batch["input_ids"], fb, max_length=8, eos_token_id=-1).clone()
    assert seen == {"mode": "thread_local", "calls": 1}, seen          # one attempt per shape state, then the state stays on enqueue
    assert any("capture of the token-step failed" in str(x.message) for x in w)
    assert torch.equal(a, ref) and torch.equal(b, ref)
    assert next(iter(model._decode_states.values()))["graph"] is False
    # the real capture still works for a new shape state
    monkeypatch.setattr(torch.cuda, "graph", real_graph)
    c = model.greedy_generate(batch["input_ids"], fb, max_length=6, eos_token_id=-1)
    assert torch.equal(c, ref[:, :6])
