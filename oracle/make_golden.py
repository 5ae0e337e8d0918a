#!/usr/bin/env python3
"""Generate tests/golden/*.npz -- golden input/output vectors for the hot path.

Runs ONLY in the build container (needs /root/reference and transformers 5.15); the GPU box
never runs this, it only reads the committed .npz files.  Everything written is DATA: seeded
inputs, explicit weights and the outputs the reference code / the library produced for them.

Sources of the expected values:
  G1  reference `VisualEmbedding`                    (VL-T5/src/modeling_t5_our.py:27-143)
  G2  reference `VLT5.calculate_current_prototype`, `update_prototype`,
      `cosine_similarity_multi` as unbound functions (:434-511) and nextqa `memory_loss`
      (VL-T5/nextqa/modeling_t5_nextqa.py:544-555), scripted 5-step task sequence
  G3  transformers-5.15 T5 leaf modules (T5LayerNorm, T5Attention enc/dec/cross, T5LayerFF relu
      and gated-gelu, encoder T5Block, 2-layer decoder T5Stack): outputs and gradients
  G4  integer tables: `_relative_position_bucket` (bidirectional / causal), `_shift_right`
  G5  loss reduction of vqa_model.py:46-54, expected values by an independent python loop
  G6  tiny full model through the restatement (regression fixture for the composition)

Usage:  python oracle/make_golden.py            (writes tests/golden/)
"""
import os
import sys
import types
import math

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference/VL-T5"
sys.path.insert(0, ROOT)


def import_reference():
    """Import the reference model file under transformers 5.15 (4 names it imports were removed)."""
    import transformers
    import transformers.models.t5.modeling_t5  # noqa: F401  (swaps the lazy module object)
    import transformers.modeling_utils as mu
    for name in ("find_pruneable_heads_and_indices", "prune_linear_layer"):
        if not hasattr(mu, name):
            setattr(mu, name, lambda *a, **k: None)
    tmod = sys.modules["transformers"]
    for name in ("BeamScorer", "BeamSearchScorer"):
        if not hasattr(tmod, name):
            try:
                setattr(tmod, name, type(name, (), {}))
            except Exception:
                tmod.__dict__[name] = type(name, (), {})
    sys.path.insert(0, os.path.join(REF, "src"))
    import importlib
    ref = importlib.import_module("modeling_t5_our")
    sys.path.pop(0)
    spec = importlib.util.spec_from_file_location("modeling_t5_nextqa",
                                                  os.path.join(REF, "nextqa", "modeling_t5_nextqa.py"))
    nq = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(nq)
    return ref, nq


def npz(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **conv)
    print(f"wrote {name}.npz  ({sum(a.nbytes for a in conv.values()) / 1e3:.1f} kB raw)")


def f16exact(t):
    """Round to fp16-representable values (still stored as f32) so fixtures compress well."""
    return t.half().float()


# ------------------------------------------------------------------------------------------
def g1_visual_embedding(ref):
    for tag, d, fd, vocab, B, N in (("tiny", 64, 64, 400, 2, 36), ("mid", 128, 256, 512, 2, 36)):
        g = torch.Generator().manual_seed(101)
        cfg = types.SimpleNamespace(feat_dim=fd, pos_dim=4, n_images=2, d_model=d, layer_norm_epsilon=1e-6,
                                    individual_vis_layer_norm=True, use_vis_layer_norm=True,
                                    use_vis_order_embedding=True)
        shared = nn.Embedding(vocab, d)
        ve = ref.VisualEmbedding(cfg, shared)
        with torch.no_grad():
            for p in list(ve.parameters()):
                p.copy_(f16exact(torch.randn(p.shape, generator=g) * (0.1 if p.dim() > 1 else 1.0)))
            shared.weight.copy_(f16exact(torch.randn(shared.weight.shape, generator=g)))
        feats = f16exact(torch.relu(torch.randn(B, N, fd, generator=g)) * 1.5)
        xs = torch.rand(B, N, 2, generator=g).sort(dim=2).values
        ys = torch.rand(B, N, 2, generator=g).sort(dim=2).values
        boxes = f16exact(torch.stack([xs[..., 0], ys[..., 0], xs[..., 1], ys[..., 1]], dim=2))
        feats.requires_grad_(True)
        out = ve(feats, boxes)
        gout = f16exact(torch.randn(out.shape, generator=g))
        out.backward(gout)
        sd = {k.replace(".", "__"): v for k, v in ve.state_dict().items() if "obj_order" not in k}
        grads = {"grad__" + k.replace(".", "__"): p.grad for k, p in ve.named_parameters()
                 if "obj_order" not in k}
        npz(f"g1_visual_embedding_{tag}", feats=feats, boxes=boxes, shared=shared.weight, out=out,
            gout=gout, grad_feats=feats.grad, grad_shared=shared.weight.grad, **sd, **grads)


def g2_prototype(ref, nq):
    d, B, S, CQ, CV = 768, 4, 24, 10, 80      # the reference hard-codes 768 (modeling_t5_our.py:503)
    g = torch.Generator().manual_seed(202)
    ns = types.SimpleNamespace(Q_task_mem_proto={}, V_task_mem_proto={}, Q_task_cur_proto={},
                               V_task_cur_proto={}, Q_prototype_num={}, V_prototype_num={}, L=20)
    tasks = [0, 0, 1, 1, 1, 2, 2]
    # batches 3 and 4 of task 1 are "rehearsal-like": they hold samples of the older task 0 too
    ques_ids = [[0, 0, 0, 0], [0, 0, 0, 0], [1, 1, 1, 1], [1, 0, 1, 0], [0, 0, 1, 1], [2, 2, 2, 2], [0, 1, 2, 2]]
    real_device = torch.device
    rec = {}
    try:
        ref.torch.device = lambda *a, **k: real_device("cpu")     # the function hard-codes 'cuda'
        for step, task in enumerate(tasks):
            hidden = f16exact(torch.randn(B, S, d, generator=g))
            ql = torch.zeros(B, CQ).scatter_(1, torch.tensor(ques_ids[step])[:, None], 1.0)
            cids = torch.randint(0, 16, (B,), generator=g)
            cl = torch.zeros(B, CV).scatter_(1, cids[:, None], 1.0)
            cq, nqn = ref.VLT5.calculate_current_prototype(ns, hidden[:, :20], ql)
            cv, nvn = ref.VLT5.calculate_current_prototype(ns, hidden[:, 20:], cl)
            # snapshot now: update_prototype later writes rows in place into tensors aliasing these
            cq_snap, cv_snap = cq.clone(), cv.clone()
            if step > 0:
                lq, lv = nq.VLT5.memory_loss(ns, hidden[:, :20], hidden[:, 20:], ql, cl)
                rec[f"s{step}_memloss"] = torch.stack([lq, lv])
            ref.VLT5.update_prototype(ns, cq, cv, nqn, nvn, task, 0.5, 0.3)
            rq, iq, _ = ref.VLT5.cosine_similarity_multi(ns, ns.Q_prototype, hidden[:, :20].mean(1), ql)
            rv, iv, _ = ref.VLT5.cosine_similarity_multi(ns, ns.V_prototype, hidden[:, 20:].mean(1), cl)
            rec.update({f"s{step}_hidden": hidden, f"s{step}_ques": ql, f"s{step}_cate": cl,
                        f"s{step}_curQ": cq_snap, f"s{step}_curV": cv_snap, f"s{step}_numQ": nqn.clone(), f"s{step}_numV": nvn.clone(),
                        f"s{step}_Qproto": ns.Q_prototype.clone(), f"s{step}_Vproto": ns.V_prototype.clone(),
                        f"s{step}_Qnum": ns.Q_prototype_num.clone(), f"s{step}_Vnum": ns.V_prototype_num.clone(),
                        f"s{step}_idxQ": iq, f"s{step}_idxV": iv, f"s{step}_retQ": rq, f"s{step}_retV": rv})
    finally:
        ref.torch.device = real_device
    # margin report: top-2 similarity gap per retrieval, so GPU tests know which argmax are "safe"
    npz("g2_prototype_sequence", tasks=np.array(tasks), alpha=np.float32(0.5), beta=np.float32(0.3), **rec)


def hf_cfg(is_decoder=False, gated=False, layers=2):
    from transformers import T5Config
    c = T5Config(vocab_size=400, d_model=64, d_kv=16, d_ff=128, num_layers=layers, num_decoder_layers=layers,
                 num_heads=4, dropout_rate=0.0, feed_forward_proj="gated-gelu" if gated else "relu",
                 is_decoder=is_decoder, is_encoder_decoder=False, use_cache=False)
    c._attn_implementation = "eager"
    return c


def seeded_fill(module, g, scale=0.2):
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.dim() == 1:
                p.copy_(f16exact(1.0 + 0.3 * torch.randn(p.shape, generator=g)))
            else:
                p.copy_(f16exact(torch.randn(p.shape, generator=g) * scale))


def g3_hf_leaves():
    from transformers.models.t5 import modeling_t5 as m
    g = torch.Generator().manual_seed(303)
    B, S, L, T, Sx, d, H = 2, 12, 7, 5, 14, 64, 4
    rec = {}
    # ---- layernorm
    ln = m.T5LayerNorm(d, eps=1e-6)
    seeded_fill(ln, g)
    x = f16exact(torch.randn(B, S, d, generator=g) * 3).requires_grad_(True)
    y = ln(x)
    gy = f16exact(torch.randn(y.shape, generator=g))
    y.backward(gy)
    rec.update(ln_x=x, ln_w=ln.weight, ln_y=y, ln_gy=gy, ln_gx=x.grad, ln_gw=ln.weight.grad)

    # ---- encoder self-attention with an explicit [B,H,S,S] bias+mask (the reference folds both)
    enc = hf_cfg(False)
    att = m.T5Attention(enc, has_relative_attention_bias=True, layer_idx=0)
    seeded_fill(att, g)
    x = f16exact(torch.randn(B, S, d, generator=g)).requires_grad_(True)
    keymask = torch.ones(B, S)
    keymask[0, 5:L] = 0
    keymask[1, 3:L] = 0
    bias = torch.zeros(1, H, S, S)
    bias[:, :, :L, :L] = att.compute_bias(L, L)
    bias = bias + (1.0 - keymask)[:, None, None, :] * -10000.0
    y = att(x, mask=None, position_bias=bias)[0]
    gy = f16exact(torch.randn(y.shape, generator=g))
    y.backward(gy)
    rec.update(ea_x=x, ea_keymask=keymask, ea_L=np.int64(L), ea_y=y, ea_gy=gy, ea_gx=x.grad,
               ea_bias=bias.detach(),
               **{f"ea_{n}": getattr(att, n).weight for n in "qkvo"},
               **{f"ea_g{n}": getattr(att, n).weight.grad for n in "qkvo"},
               ea_rel=att.relative_attention_bias.weight, ea_grel=att.relative_attention_bias.weight.grad)

    # ---- decoder causal self-attention (unidirectional buckets) and cross-attention
    dec = hf_cfg(True)
    att = m.T5Attention(dec, has_relative_attention_bias=True, layer_idx=0, is_causal=True)
    seeded_fill(att, g)
    x = f16exact(torch.randn(B, T, d, generator=g)).requires_grad_(True)
    causal = torch.tril(torch.ones(T, T))
    bias = att.compute_bias(T, T) + (1.0 - causal)[None, None] * -10000.0
    y = att(x, mask=None, position_bias=bias)[0]
    gy = f16exact(torch.randn(y.shape, generator=g))
    y.backward(gy)
    rec.update(da_x=x, da_y=y, da_gy=gy, da_gx=x.grad,
               **{f"da_{n}": getattr(att, n).weight for n in "qkvo"},
               **{f"da_g{n}": getattr(att, n).weight.grad for n in "qkvo"},
               da_rel=att.relative_attention_bias.weight, da_grel=att.relative_attention_bias.weight.grad)

    att = m.T5Attention(dec, has_relative_attention_bias=False, layer_idx=0)
    seeded_fill(att, g)
    x = f16exact(torch.randn(B, T, d, generator=g)).requires_grad_(True)
    mem = f16exact(torch.randn(B, Sx, d, generator=g)).requires_grad_(True)
    kmask = torch.ones(B, Sx)
    kmask[0, 4:6] = 0
    bias = ((1.0 - kmask)[:, None, None, :] * -1e9).expand(B, 1, T, Sx)
    y = att(x, mask=None, key_value_states=mem, position_bias=bias)[0]
    gy = f16exact(torch.randn(y.shape, generator=g))
    y.backward(gy)
    rec.update(ca_x=x, ca_mem=mem, ca_kmask=kmask, ca_y=y, ca_gy=gy, ca_gx=x.grad, ca_gmem=mem.grad,
               **{f"ca_{n}": getattr(att, n).weight for n in "qkvo"},
               **{f"ca_g{n}": getattr(att, n).weight.grad for n in "qkvo"})

    # ---- FFN layer (LN + dense + residual), relu and gated-gelu
    for tag, gated in (("ff", False), ("gff", True)):
        ff = m.T5LayerFF(hf_cfg(False, gated))
        seeded_fill(ff, g)
        x = f16exact(torch.randn(B, S, d, generator=g)).requires_grad_(True)
        y = ff(x)
        gy = f16exact(torch.randn(y.shape, generator=g))
        y.backward(gy)
        rec.update({f"{tag}_x": x, f"{tag}_y": y, f"{tag}_gy": gy, f"{tag}_gx": x.grad})
        for n, p in ff.named_parameters():
            rec[f"{tag}_{n.replace('.', '__')}"] = p
            rec[f"{tag}_g__{n.replace('.', '__')}"] = p.grad
    npz("g3_hf_leaves", **rec)

    # ---- one encoder T5Block and a 2-layer decoder T5Stack
    rec = {}
    blk = m.T5Block(enc, has_relative_attention_bias=True, layer_idx=0)
    seeded_fill(blk, g)
    x = f16exact(torch.randn(B, S, d, generator=g)).requires_grad_(True)
    keymask = torch.ones(B, S)
    keymask[1, 2:L] = 0
    bias = torch.zeros(1, H, S, S)
    bias[:, :, :L, :L] = blk.layer[0].SelfAttention.compute_bias(L, L)
    bias = bias + (1.0 - keymask)[:, None, None, :] * -10000.0
    y = blk(x, attention_mask=None, position_bias=bias)[0]
    gy = f16exact(torch.randn(y.shape, generator=g))
    y.backward(gy)
    rec.update(eb_x=x, eb_keymask=keymask, eb_L=np.int64(L), eb_y=y, eb_gy=gy, eb_gx=x.grad)
    for n, p in blk.named_parameters():
        rec["eb_p__" + n.replace(".", "__")] = p
        rec["eb_g__" + n.replace(".", "__")] = p.grad

    stack = m.T5Stack(dec)
    seeded_fill(stack, g)
    ids = torch.randint(0, 400, (B, T), generator=g)
    mem = f16exact(torch.randn(B, Sx, d, generator=g)).requires_grad_(True)
    kmask = torch.ones(B, Sx)
    kmask[1, 3:7] = 0
    y = stack(input_ids=ids, encoder_hidden_states=mem, encoder_attention_mask=kmask).last_hidden_state
    gy = f16exact(torch.randn(y.shape, generator=g))
    y.backward(gy)
    rec.update(ds_ids=ids, ds_mem=mem, ds_kmask=kmask, ds_y=y, ds_gy=gy, ds_gmem=mem.grad)
    for n, p in stack.named_parameters():
        rec["ds_p__" + n.replace(".", "__")] = p
        rec["ds_g__" + n.replace(".", "__")] = p.grad
    npz("g3_hf_block_stack", **rec)


def g4_integer_tables():
    from transformers.models.t5 import modeling_t5 as m
    from transformers import T5Config
    q = torch.arange(160)[:, None]
    k = torch.arange(160)[None, :]
    bi = m.T5Attention._relative_position_bucket(k - q, bidirectional=True, num_buckets=32, max_distance=128)
    uni = m.T5Attention._relative_position_bucket(k - q, bidirectional=False, num_buckets=32, max_distance=128)
    cfg = T5Config(vocab_size=400, d_model=64, d_kv=16, d_ff=128, num_layers=1, num_heads=4,
                   decoder_start_token_id=0, pad_token_id=0)
    model = m.T5ForConditionalGeneration(cfg)
    labels = torch.tensor([[5, 9, 1, -100, -100], [7, 1, -100, -100, -100], [3, 4, 5, 6, 1], [-100] * 5])
    npz("g4_integer_tables", bucket_bidirectional=bi, bucket_causal=uni, labels=labels,
        shifted=model._shift_right(labels))


def g5_loss_reduction():
    g = torch.Generator().manual_seed(505)
    B, T = 6, 5
    labels = torch.randint(2, 300, (B, T), generator=g)
    lens = [5, 3, 1, 0, 2, 4]                      # row 3 is all padding
    for b, n in enumerate(lens):
        labels[b, n:] = -100
    tok = torch.rand(B * T, generator=g) * 4
    tok = tok * (labels.reshape(-1) != -100)       # CE(reduction='none', ignore_index) gives 0 there
    scores = torch.tensor([1.0, 0.6, 0.0, 0.9, 0.3, 1.0])
    acc = 0.0
    for b in range(B):
        s, n = 0.0, 0
        for t in range(T):
            if labels[b, t].item() != -100:
                s += float(tok[b * T + t])
                n += 1
        acc += (s / max(n, 1)) * float(scores[b])
    npz("g5_loss_reduction", labels=labels, loss_tok=tok, scores=scores, expected=np.float64(acc / B))


def g6_tiny_model():
    from oracle import ref_cpu as R
    cfg = R.tiny_cfg()
    P = {k: f16exact(v) for k, v in R.init_params(cfg, seed=606).items()}
    # break the "all norms == 1" symmetry so norm-weight gradients are exercised
    g = torch.Generator().manual_seed(607)
    for k in P:
        if P[k].dim() == 1 and not k.endswith("bias"):
            P[k] = f16exact(1.0 + 0.2 * torch.randn(P[k].shape, generator=g))
    model = R.OracleModel(cfg, P)
    rec = {"p__" + k.replace(".", "__"): v for k, v in P.items()}
    for step, (task, seed) in enumerate(((0, 11), (0, 12), (1, 13))):
        batch = R.synthetic_batch(cfg, B=4, L=9 if step == 1 else 12, V=36, T=5, seed=seed, task_id=task)
        batch["vis_feats"] = f16exact(batch["vis_feats"])
        batch["boxes"] = f16exact(batch["boxes"])
        model.zero_grad()
        out = model.train_step(batch, task, 0.5, 0.3, training=True)   # dropout = 0 in tiny_cfg
        out["loss"].backward()
        for k, v in batch.items():
            rec[f"s{step}_in_{k}"] = v
        rec.update({f"s{step}_logits": out["logits"], f"s{step}_loss": out["loss"], f"s{step}_loss_tok": out["loss_tok"],
                    f"s{step}_enc": out["encoder_hidden_states"], f"s{step}_idxQ": out["max_idx_Q"],
                    f"s{step}_idxV": out["max_idx_V"], f"s{step}_Qproto": model.state.Q_prototype.clone(),
                    f"s{step}_Vproto": model.state.V_prototype.clone()})
        if step == 0:
            for k, p in model.P.items():
                if p.grad is not None:
                    rec["s0_g__" + k.replace(".", "__")] = p.grad.clone()
    npz("g6_tiny_model", **rec)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(4)
    ref, nq = import_reference()
    g1_visual_embedding(ref)
    g2_prototype(ref, nq)
    g3_hf_leaves()
    g4_integer_tables()
    g5_loss_reduction()
    g6_tiny_model()
