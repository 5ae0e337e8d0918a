"""CPU tests: the oracle (oracle/ref_cpu.py) against the golden vectors in tests/golden/.

The golden vectors were produced by oracle/make_golden.py from the reference's own importable
code (VisualEmbedding, prototype-head methods, memory_loss) and from the transformers-5.15 T5
leaf modules; these tests pin the restatement to them.
"""
import torch
import pytest

from oracle import ref_cpu as R
from conftest import load_golden

TOL = dict(rtol=2e-5, atol=2e-5)


def un(d, prefix):
    return {k[len(prefix):].replace("__", "."): v for k, v in d.items() if k.startswith(prefix)}


@pytest.mark.parametrize("tag,d,fd,vocab", [("tiny", 64, 64, 400), ("mid", 128, 256, 512)])
def test_g1_visual_embedding(tag, d, fd, vocab):
    G = load_golden(f"g1_visual_embedding_{tag}")
    cfg = R.Cfg(d_model=d, feat_dim=fd, vocab_size=vocab)
    pre = "encoder.visual_embedding."
    P = {"shared.weight": G["shared"].clone().requires_grad_(True)}
    for k in ("feat_embedding.0.weight", "feat_embedding.0.bias", "feat_embedding.1.weight",
              "absolute_vis_pos_embedding.0.weight", "absolute_vis_pos_embedding.0.bias",
              "absolute_vis_pos_embedding.1.weight", "img_order_embedding.weight"):
        P[pre + k] = G[k.replace(".", "__")].clone().requires_grad_(True)
    feats = G["feats"].clone().requires_grad_(True)
    out = R.visual_embedding(P, feats, G["boxes"], cfg)
    torch.testing.assert_close(out, G["out"], **TOL)
    out.backward(G["gout"])
    torch.testing.assert_close(feats.grad, G["grad_feats"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(P["shared.weight"].grad, G["grad_shared"], **TOL)
    for k in ("feat_embedding.0.weight", "feat_embedding.1.weight", "absolute_vis_pos_embedding.0.weight",
              "absolute_vis_pos_embedding.0.bias", "img_order_embedding.weight"):
        torch.testing.assert_close(P[pre + k].grad, G["grad__" + k.replace(".", "__")], rtol=1e-4, atol=1e-4)


def test_g2_prototype_sequence():
    G = load_golden("g2_prototype_sequence")
    st = R.PrototypeState()
    alpha, beta = float(G["alpha"]), float(G["beta"])
    for step, task in enumerate(G["tasks"].tolist()):
        h = G[f"s{step}_hidden"]
        ql, cl = G[f"s{step}_ques"], G[f"s{step}_cate"]
        cq, nq = R.calculate_current_prototype(h[:, :20], ql)
        cv, nv = R.calculate_current_prototype(h[:, 20:], cl)
        torch.testing.assert_close(cq, G[f"s{step}_curQ"], rtol=0, atol=0)
        torch.testing.assert_close(cv, G[f"s{step}_curV"], rtol=0, atol=0)
        assert torch.equal(nq, G[f"s{step}_numQ"]) and torch.equal(nv, G[f"s{step}_numV"])
        if step > 0:
            lq, lv = R.memory_loss(h[:, :20], h[:, 20:], ql, cl, st.Q_prototype, st.V_prototype)
            torch.testing.assert_close(torch.stack([lq, lv]), G[f"s{step}_memloss"], rtol=1e-6, atol=0)
        st.update(cq, cv, nq, nv, task, alpha, beta)
        torch.testing.assert_close(st.Q_prototype, G[f"s{step}_Qproto"], rtol=0, atol=0)
        torch.testing.assert_close(st.V_prototype, G[f"s{step}_Vproto"], rtol=0, atol=0)
        assert torch.equal(st.Q_prototype_num, G[f"s{step}_Qnum"])
        assert torch.equal(st.V_prototype_num, G[f"s{step}_Vnum"])
        rq, iq = R.cosine_retrieve(st.Q_prototype, h[:, :20].mean(1))
        rv, iv = R.cosine_retrieve(st.V_prototype, h[:, 20:].mean(1))
        assert torch.equal(iq, G[f"s{step}_idxQ"]), "integer prototype indices must be bit-exact"
        assert torch.equal(iv, G[f"s{step}_idxV"])
        torch.testing.assert_close(rq, G[f"s{step}_retQ"], rtol=0, atol=0)
        torch.testing.assert_close(rv, G[f"s{step}_retV"], rtol=0, atol=0)


def tiny():
    return R.tiny_cfg()


def test_g3_layernorm():
    G = load_golden("g3_hf_leaves")
    x = G["ln_x"].clone().requires_grad_(True)
    w = G["ln_w"].clone().requires_grad_(True)
    y = R.t5_layernorm(x, w, 1e-6)
    torch.testing.assert_close(y, G["ln_y"], **TOL)
    y.backward(G["ln_gy"])
    torch.testing.assert_close(x.grad, G["ln_gx"], **TOL)
    torch.testing.assert_close(w.grad, G["ln_gw"], rtol=1e-4, atol=1e-4)


def _attn_case(G, p, bias, kv=None):
    cfg = tiny()
    x = G[p + "_x"].clone().requires_grad_(True)
    W = {n: G[f"{p}_{n}"].clone().requires_grad_(True) for n in "qkvo"}
    mem = None if kv is None else G[kv].clone().requires_grad_(True)
    y = R.t5_attention(x, x if mem is None else mem, W["q"], W["k"], W["v"], W["o"], bias, cfg, 0.0, False)
    torch.testing.assert_close(y, G[p + "_y"], **TOL)
    y.backward(G[p + "_gy"])
    torch.testing.assert_close(x.grad, G[p + "_gx"], rtol=1e-4, atol=1e-4)
    for n in "qkvo":
        torch.testing.assert_close(W[n].grad, G[f"{p}_g{n}"], rtol=1e-4, atol=1e-4)
    return mem


def test_g3_encoder_attention_bias_and_mask():
    G = load_golden("g3_hf_leaves")
    cfg = tiny()
    L = int(G["ea_L"])
    S = G["ea_x"].shape[1]
    rel = G["ea_rel"].clone().requires_grad_(True)
    bias = torch.zeros(1, cfg.num_heads, S, S)
    bias[:, :, :L, :L] = R.compute_bias(rel, L, L, True, cfg)
    bias = bias + (1.0 - G["ea_keymask"])[:, None, None, :] * -10000.0
    torch.testing.assert_close(bias, G["ea_bias"], rtol=0, atol=0)
    _attn_case(G, "ea", bias)
    torch.testing.assert_close(rel.grad, G["ea_grel"], rtol=1e-4, atol=1e-4)


def test_g3_decoder_causal_attention():
    G = load_golden("g3_hf_leaves")
    cfg = tiny()
    T = G["da_x"].shape[1]
    rel = G["da_rel"].clone().requires_grad_(True)
    bias = R.compute_bias(rel, T, T, False, cfg) + (1.0 - torch.tril(torch.ones(T, T)))[None, None] * -10000.0
    _attn_case(G, "da", bias)
    torch.testing.assert_close(rel.grad, G["da_grel"], rtol=1e-4, atol=1e-4)


def test_g3_cross_attention():
    G = load_golden("g3_hf_leaves")
    B, T = G["ca_x"].shape[:2]
    bias = ((1.0 - G["ca_kmask"])[:, None, None, :] * -1e9).expand(B, 1, T, G["ca_kmask"].shape[1])
    mem = _attn_case(G, "ca", bias, kv="ca_mem")
    torch.testing.assert_close(mem.grad, G["ca_gmem"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tag,gated", [("ff", False), ("gff", True)])
def test_g3_ffn_layer(tag, gated):
    G = load_golden("g3_hf_leaves")
    cfg = R.tiny_cfg(gated_act=gated)
    P = {}
    for k, v in G.items():
        if k.startswith(tag + "_") and "__" in k and not k.startswith(tag + "_g__"):
            P[k[len(tag) + 1:].replace("__", ".")] = v.clone().requires_grad_(True)
    x = G[tag + "_x"].clone().requires_grad_(True)
    xn = R.t5_layernorm(x, P["layer_norm.weight"], cfg.eps)
    y = x + R.t5_ffn(xn, P, "DenseReluDense.", cfg, False)
    torch.testing.assert_close(y, G[tag + "_y"], **TOL)
    y.backward(G[tag + "_gy"])
    torch.testing.assert_close(x.grad, G[tag + "_gx"], rtol=1e-4, atol=1e-4)
    for k, p in P.items():
        torch.testing.assert_close(p.grad, G[f"{tag}_g__" + k.replace(".", "__")], rtol=1e-4, atol=1e-4)


def test_g3_encoder_block():
    """One full encoder block (self-attention sublayer + FFN sublayer) against T5Block."""
    G = load_golden("g3_hf_block_stack")
    cfg = R.tiny_cfg(num_layers=1)
    Praw = un(G, "eb_p__")
    P = {"encoder.block.0." + k: v.clone().requires_grad_(True) for k, v in Praw.items()}
    x = G["eb_x"].clone().requires_grad_(True)
    L, S = int(G["eb_L"]), x.shape[1]
    pre = "encoder.block.0.layer."
    a = pre + "0.SelfAttention."
    bias = torch.zeros(1, cfg.num_heads, S, S)
    bias[:, :, :L, :L] = R.compute_bias(P[a + "relative_attention_bias.weight"], L, L, True, cfg)
    bias = bias + (1.0 - G["eb_keymask"])[:, None, None, :] * -10000.0
    xn = R.t5_layernorm(x, P[pre + "0.layer_norm.weight"], cfg.eps)
    h = x + R.t5_attention(xn, xn, P[a + "q.weight"], P[a + "k.weight"], P[a + "v.weight"], P[a + "o.weight"],
                           bias, cfg, 0.0, False)
    hn = R.t5_layernorm(h, P[pre + "1.layer_norm.weight"], cfg.eps)
    y = h + R.t5_ffn(hn, P, pre + "1.DenseReluDense.", cfg, False)
    torch.testing.assert_close(y, G["eb_y"], **TOL)
    y.backward(G["eb_gy"])
    torch.testing.assert_close(x.grad, G["eb_gx"], rtol=1e-4, atol=1e-4)
    for k in Praw:
        torch.testing.assert_close(P["encoder.block.0." + k].grad, G["eb_g__" + k.replace(".", "__")],
                                   rtol=2e-4, atol=2e-4)


def test_g3_decoder_stack():
    """The oracle's decoder (embedding, causal + cross attention, FFN, final norm) against T5Stack."""
    G = load_golden("g3_hf_block_stack")
    cfg = R.tiny_cfg()
    Praw = un(G, "ds_p__")
    P = {}
    for k, v in Praw.items():
        name = "shared.weight" if k == "embed_tokens.weight" else "decoder." + k
        P[name] = v.clone().requires_grad_(True)
    mem = G["ds_mem"].clone().requires_grad_(True)
    y = R.decoder_forward(P, G["ds_ids"], mem, G["ds_kmask"], cfg, False)
    torch.testing.assert_close(y, G["ds_y"], **TOL)
    y.backward(G["ds_gy"])
    torch.testing.assert_close(mem.grad, G["ds_gmem"], rtol=1e-4, atol=1e-4)
    for k in Praw:
        name = "shared.weight" if k == "embed_tokens.weight" else "decoder." + k
        torch.testing.assert_close(P[name].grad, G["ds_g__" + k.replace(".", "__")], rtol=2e-4, atol=2e-4)


def test_g4_integer_tables_bit_exact():
    G = load_golden("g4_integer_tables")
    n = G["bucket_bidirectional"].shape[0]
    assert torch.equal(R.bucket_table(n, n, True), G["bucket_bidirectional"])
    assert torch.equal(R.bucket_table(n, n, False), G["bucket_causal"])
    assert torch.equal(R.shift_right(G["labels"], R.Cfg()), G["shifted"])


def test_g5_loss_reduction_known_answer():
    G = load_golden("g5_loss_reduction")
    got = R.train_step_loss(G["loss_tok"], G["labels"], G["scores"])
    assert abs(float(got) - float(G["expected"])) < 1e-6


def test_g6_tiny_model_regression():
    """The composed model reproduces its committed fixture (guards against oracle drift)."""
    G = load_golden("g6_tiny_model")
    cfg = R.tiny_cfg()
    model = R.OracleModel(cfg, un(G, "p__"))
    for step, task in enumerate((0, 0, 1)):
        batch = {k[len(f"s{step}_in_"):]: v for k, v in G.items() if k.startswith(f"s{step}_in_")}
        model.zero_grad()
        out = model.train_step(batch, task, 0.5, 0.3, training=True)
        torch.testing.assert_close(out["logits"], G[f"s{step}_logits"], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(out["loss"], G[f"s{step}_loss"], rtol=1e-5, atol=1e-5)
        assert torch.equal(out["max_idx_Q"], G[f"s{step}_idxQ"])
        assert torch.equal(out["max_idx_V"], G[f"s{step}_idxV"])
        torch.testing.assert_close(model.state.Q_prototype, G[f"s{step}_Qproto"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(model.state.V_prototype, G[f"s{step}_Vproto"], rtol=1e-4, atol=1e-5)
        if step == 0:
            out["loss"].backward()
            for k, g in un(G, "s0_g__").items():
                torch.testing.assert_close(model.P[k].grad, g, rtol=1e-3, atol=1e-5)


def test_optimizer_restatement_matches_torch_adamw_closely():
    """HF-AdamW restatement vs torch.optim.AdamW: same to ~1e-6 (they differ only in where eps and
    the decay sit), which guards against gross errors in the restatement."""
    torch.manual_seed(0)
    p0 = torch.randn(50, 7)
    a = {"w.weight": p0.clone().requires_grad_(True)}
    b = torch.nn.Parameter(p0.clone())
    oa = R.HFAdamW(a, lr=1e-3, eps=1e-6, weight_decay=0.01)
    ob = torch.optim.AdamW([b], lr=1e-3, eps=1e-6, weight_decay=0.01, betas=(0.9, 0.999))
    for i in range(5):
        g = torch.randn(50, 7)
        a["w.weight"].grad = g.clone()
        b.grad = g.clone()
        oa.step()
        ob.step()
    torch.testing.assert_close(a["w.weight"].detach(), b.detach(), rtol=1e-4, atol=2e-5)


def test_weight_decay_grouping_quirk():
    assert R.weight_decay_of("encoder.block.3.layer.0.layer_norm.weight", 0.01) == 0.01   # T5 norms DO decay
    assert R.weight_decay_of("encoder.visual_embedding.feat_embedding.0.bias", 0.01) == 0.0
    assert R.weight_decay_of("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", 0.01) == 0.0
