"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, the parameter layout
mirrors the reference's state_dict, the bucket tables are bit-exact, the module tree is drop-in.  No compute calls."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden, ROOT


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    from vqacl_amd import _lib
    return _lib


def test_library_exports_every_symbol_of_the_header(built):
    hdr = open(os.path.join(ROOT, "include", "vlt5_hip.h")).read()
    declared = set(re.findall(r"\b(vlt5_[a-z0-9_]+)\s*\(", hdr))
    raw = C.CDLL(built.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"libvlt5_hip.so does not export {name}"
    assert set(built.PROTOTYPES) == declared, (set(built.PROTOTYPES) ^ declared)
    assert built.lib().vlt5_abi_version() == 8 and built.lib().vlt5_build_flags() == 0


def test_experiment_builds_are_refused_by_the_loader(built, tmp_path, monkeypatch):
    """Round-5 advisor: libraries compiled with -DENC_DGRAD_HOT_A / -DATTN_BWD_NO_STORE (upper-bound measurements, results wrong on purpose)
    carried the product's ABI number and loaded through VLT5_LIB without complaint.  They now report vlt5_build_flags() != 0 and the loader
    refuses them unless VLT5_ALLOW_EXPERIMENT=1.  (A stand-in library with the two entry points is enough to exercise the loader.)"""
    import subprocess
    src = tmp_path / "fake.c"
    src.write_text("int vlt5_abi_version(void){return %d;}\nint vlt5_build_flags(void){return 3;}\n" % built.ABI_VERSION)
    so = tmp_path / "libfake.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)])
    code = ("import sys; sys.path.insert(0, %r)\nfrom vqacl_amd import _lib as L\nL.PROTOTYPES = {k: v for k, v in L.PROTOTYPES.items() "
            "if k in ('vlt5_abi_version', 'vlt5_build_flags')}\n"
            "try:\n    L.lib(); print('LOADED')\nexcept L.Vlt5Error as e:\n    print('REFUSED', e)\n" % ROOT)
    env = dict(os.environ, VLT5_LIB=str(so))
    env.pop("VLT5_ALLOW_EXPERIMENT", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120).stdout
    assert "REFUSED" in out and "EXPERIMENT build" in out and "ENC_DGRAD_HOT_A" in out and "ATTN_BWD_NO_STORE" in out, out
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, VLT5_ALLOW_EXPERIMENT="1"), capture_output=True, text=True, timeout=120).stdout
    assert "LOADED" in out, out


def test_struct_sizes_match_the_c_side(built, tmp_path):
    """Compile a tiny C program against the header and compare sizeof() with the ctypes mirrors."""
    import subprocess
    src = tmp_path / "sz.c"
    names = ["vlt5_gemm_desc", "vlt5_attn_desc", "vlt5_config", "vlt5_step", "vlt5_enc_attn_desc", "vlt5_tuning", "vlt5_greedy_desc",
             "vlt5_decode_linear_desc", "vlt5_dec_attn_desc", "vlt5_stack_inputs_desc", "vlt5_proto_head_desc", "vlt5_dec_attn_grads",
             "vlt5_ffn_desc", "vlt5_ffn_grads", "vlt5_lmhead_ce_desc", "vlt5_lmhead_ce_grads"]
    mirrors = [built.GemmDesc, built.AttnDesc, built.Config, built.Step, built.EncAttnDesc, built.Tuning, built.GreedyDesc,
               built.DecodeLinearDesc, built.DecAttnDesc, built.StackInputsDesc, built.ProtoHeadDesc, built.DecAttnGrads, built.FfnDesc,
               built.FfnGrads, built.LmheadCeDesc, built.LmheadCeGrads]
    src.write_text('#include <stdio.h>\n#include "vlt5_hip.h"\nint main(){printf("' + " ".join(["%zu"] * len(names)) + '\\n", '
                   + ", ".join(f"sizeof({n})" for n in names) + ');return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(m) for m in mirrors], list(zip(names, sizes, [C.sizeof(m) for m in mirrors]))


def test_the_library_reads_no_environment_and_tuning_comes_from_the_host(built):
    """re-entrant, no global state.  The experiment switches of the engine travel in a vlt5_tuning record (vlt5_step.tuning,
    vlt5_gemm_desc.tuning); the VLT5_* variables of the A/B scripts are read once on the Python side."""
    import glob
    for f in glob.glob(os.path.join(ROOT, "vqacl_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "vqacl_amd", "csrc", "*.h")):
        assert "getenv" not in open(f).read(), f"{os.path.basename(f)} reads the environment"
    raw = C.CDLL(built.LIB_PATH)
    import subprocess
    syms = subprocess.check_output(["nm", "-D", "--undefined-only", built.LIB_PATH], text=True)
    assert " getenv" not in syms and "secure_getenv" not in syms, "libvlt5_hip.so imports getenv"
    t = built.tuning_from_env({"VLT5_FOLD_NORM": "0", "VLT5_FOLD_NORM_DEC": "1", "VLT5_ENC_CUT": "4", "VLT5_DECODE_FAST": "0", "UNRELATED": "x"})
    assert (t.fold_norm, t.fold_norm_dec, t.enc_cut, t.decode_fast, t.fused_attn, t.gemm_split_kmin) == (1, 2, 4, 1, 0, 0)
    d = built.tuning_from_env({})
    assert bytes(d) == bytes(C.sizeof(built.Tuning)), "no variable set: the all-zero record (= the library's defaults)"
    m = built.make_tuning(fold_norm=False, dec_fused=True, gemm_t256_min=160)
    assert (m.fold_norm, m.dec_fused, m.gemm_t256_min) == (1, 2, 160)
    with pytest.raises(ValueError):
        built.make_tuning(no_such_switch=1)
    assert built.lib().vlt5_encoder_late_layers_tuned(12, C.byref(t)) == 4 and built.lib().vlt5_encoder_late_layers_tuned(12, None) == 6
    # the automatic split-K rule with and without a tuned threshold (no GPU involved: pure host arithmetic)
    tk = built.make_tuning(gemm_split_kmin=4096)
    assert built.lib().vlt5_gemm_auto_split_tuned(400, 768, 768, 1 << 30, None) == built.lib().vlt5_gemm_auto_split(400, 768, 768, 1 << 30) > 1
    assert built.lib().vlt5_gemm_auto_split_tuned(400, 768, 768, 1 << 30, C.byref(tk)) == 1


def test_bucket_tables_bit_exact_vs_library_golden():
    from vqacl_amd.buckets import bucket_table
    G = load_golden("g4_integer_tables")
    n = G["bucket_bidirectional"].shape[0]
    assert np.array_equal(bucket_table(n, n, True), G["bucket_bidirectional"].numpy())
    assert np.array_equal(bucket_table(n, n, False), G["bucket_causal"].numpy())
    assert bucket_table(20, 20, True).dtype == np.int32


def test_parameter_layout_mirrors_reference_state_dict(built):
    from vqacl_amd import VLT5Config, param_layout
    from oracle import ref_cpu as R
    for ocfg in (R.tiny_cfg(), R.Cfg()):
        cfg = VLT5Config(d_model=ocfg.d_model, d_kv=ocfg.d_kv, num_heads=ocfg.num_heads, d_ff=ocfg.d_ff,
                         num_layers=ocfg.num_layers, vocab_size=ocfg.vocab_size, feat_dim=ocfg.feat_dim)
        layout, total, nb = param_layout(cfg)
        shapes = R.param_shapes(ocfg)
        assert {n for n, *_ in layout} == set(shapes)
        prev_end = 0
        for name, off, shape, bucket, decay, used in layout:
            assert tuple(shape) == tuple(shapes[name]), name
            assert off % 64 == 0 and off >= prev_end
            prev_end = off + int(np.prod(shape))
            assert decay == (R.weight_decay_of(name, 0.01) > 0), name
            assert used == (not name.startswith("prototype_fc"))
        assert prev_end <= total
        assert nb == ocfg.num_layers + ocfg.num_decoder_layers + 2
        # fused operands are adjacent: q,k,v of a self-attention, and the cross k,v of all decoder layers
        by = {n: (o, s) for n, o, s, *_ in layout}
        inner = ocfg.inner
        q, k, v = (by[f"encoder.block.0.layer.0.SelfAttention.{x}.weight"][0] for x in "qkv")
        assert k - q == inner * ocfg.d_model and v - k == inner * ocfg.d_model
        k0 = by["decoder.block.0.layer.1.EncDecAttention.k.weight"][0]
        v1 = by[f"decoder.block.{ocfg.num_decoder_layers - 1}.layer.1.EncDecAttention.v.weight"][0]
        assert v1 - k0 == (2 * ocfg.num_decoder_layers - 1) * inner * ocfg.d_model
    # base model: the 225.7 M parameters of the survey
    n_params = sum(int(np.prod(s)) for _, _, s, *_ in layout)
    assert abs(n_params - 225.72e6) < 0.05e6


def test_module_tree_is_drop_in_on_cpu(built):
    """state_dict names/aliases, named_parameters order, weight-decay grouping, init_bert_weights-style apply()."""
    from vqacl_amd import VLT5VQA, VLT5Config, reference_param_groups
    from oracle import ref_cpu as R
    ocfg = R.tiny_cfg()
    m = VLT5VQA(VLT5Config(d_model=64, d_kv=16, num_heads=4, d_ff=128, num_layers=2, vocab_size=400, feat_dim=64),
                device="cpu")
    sd = m.state_dict()
    for k, shp in R.param_shapes(ocfg).items():
        assert tuple(sd[k].shape) == tuple(shp)
    for alias in ("encoder.embed_tokens.weight", "decoder.embed_tokens.weight", "lm_head.weight",
                  "encoder.visual_embedding.obj_order_embedding.weight"):
        assert sd[alias].data_ptr() == sd["shared.weight"].data_ptr()
    names = [n for n, _ in m.named_parameters()]
    assert names[0] == "shared.weight" and len(names) == len(R.param_shapes(ocfg))
    groups = reference_param_groups(m, 0.01)
    nd = {id(p) for p in groups[1]["params"]}
    named = dict(m.named_parameters())
    assert id(named["encoder.visual_embedding.feat_embedding.0.bias"]) in nd
    assert id(named["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]) in nd
    assert id(named["encoder.block.0.layer.0.layer_norm.weight"]) not in nd          # T5 norms DO decay

    # the reference's re-initialisation walks nn.Linear / nn.Embedding modules (trainer_base.py:227-237)
    def init_bert_weights(module):
        if isinstance(module, (torch.nn.Linear, torch.nn.Embedding)):
            module.weight.data.fill_(0.5)
        if isinstance(module, torch.nn.Linear) and module.bias is not None:
            module.bias.data.zero_()
    m.apply(init_bert_weights)
    assert float(m.flat_params()[m._pinfo["shared.weight"][0]]) == 0.5, "parameters are views of the flat buffer"
    assert float(named["decoder.block.1.layer.2.DenseReluDense.wi.weight"].detach()[0, 0]) == 0.5
    # load_state_dict keeps the flat aliasing
    new = {k: torch.full_like(v, 0.25) for k, v in sd.items()}
    m.load_state_dict(new)
    assert float(m.flat_params()[m._pinfo["encoder.final_layer_norm.weight"][0]]) == 0.25
    # resize_token_embeddings keeps the old rows (vqacl.py:98-99)
    m.resize_token_embeddings(408)
    assert m.state_dict()["shared.weight"].shape == (408, 64)
    assert float(m.state_dict()["shared.weight"][0, 0]) == 0.25
    # the compute path refuses to run without a GPU -- no silent fallback
    from vqacl_amd._lib import Vlt5Error
    b = R.synthetic_batch(ocfg, B=2, L=8, V=36, T=3)
    with pytest.raises(Vlt5Error):
        m.train_step(b, 0, 0.5, 0.3)


def test_init_weights_follows_the_reference_distributions(built):
    """a-18: `TrainerBase.init_weights` (trainer_base.py:218-238: every nn.Linear / nn.Embedding weight ~ N(0,1), biases 0) followed
    by HF T5 `_init_weights` (factor 1.0): q ~ N(0,(d*d_kv)^-1/2); k, v, wi, relative_attention_bias ~ N(0,d^-1/2);
    o ~ N(0,(H*d_kv)^-1/2); wo ~ N(0,d_ff^-1/2); shared ~ N(0,1); every T5LayerNorm weight = 1.  What T5 does not touch keeps N(0,1):
    the VisualEmbedding linears, img_order_embedding, prototype_fc1/2.  Checked per tensor on the sample mean / std."""
    import math
    from vqacl_amd import VLT5VQA, VLT5Config
    torch.manual_seed(11)
    d, dkv, H, ff = 128, 32, 4, 512
    for proj in ("relu", "gated-gelu"):
        m = VLT5VQA(VLT5Config(d_model=d, d_kv=dkv, num_heads=H, d_ff=ff, num_layers=2, vocab_size=1000, feat_dim=256,
                               feed_forward_proj=proj), device="cpu")
        seen = set()
        for name, p in m.named_parameters():
            v = p.detach().float()
            if name.endswith("layer_norm.weight") or name.endswith("feat_embedding.1.weight") or name.endswith("absolute_vis_pos_embedding.1.weight"):
                assert bool((v == 1).all()), name
                seen.add("norm")
                continue
            if name.endswith(".bias"):
                assert bool((v == 0).all()), name
                seen.add("bias")
                continue
            if ".SelfAttention.q." in name or ".EncDecAttention.q." in name:
                kind, std = "q", (d * dkv) ** -0.5
            elif "Attention.k." in name or "Attention.v." in name:
                kind, std = "kv", d ** -0.5
            elif "relative_attention_bias" in name:
                kind, std = "rel", d ** -0.5
            elif "Attention.o." in name:
                kind, std = "o", (H * dkv) ** -0.5
            elif ".wi." in name or ".wi_0." in name or ".wi_1." in name:
                kind, std = "wi", d ** -0.5
            elif ".wo." in name:
                kind, std = "wo", ff ** -0.5
            else:
                kind, std = "unit:" + name.split(".")[-2], 1.0       # shared, visual-embedding linears, img_order, prototype_fc
            n = v.numel()
            tol = 6.0 / math.sqrt(2 * n)                               # six standard errors of a sample standard deviation
            assert abs(float(v.std()) / std - 1.0) < tol, (name, float(v.std()), std)
            assert abs(float(v.mean())) < 6.0 * std / math.sqrt(n), (name, float(v.mean()))
            seen.add(kind)
        expect = {"norm", "bias", "q", "kv", "rel", "o", "wi", "wo", "unit:shared", "unit:0", "unit:img_order_embedding",
                  "unit:prototype_fc1", "unit:prototype_fc2"}
        assert expect <= seen, expect - seen
        wi_names = [n for n, _ in m.named_parameters() if "DenseReluDense.wi" in n]
        assert all((".wi_0." in n or ".wi_1." in n) == (proj == "gated-gelu") for n in wi_names)


def test_workspace_plan_is_consistent(built):
    from vqacl_amd import VLT5Config
    cfg = VLT5Config()
    c = cfg.c_struct()
    L = built.lib()
    need = L.vlt5_workspace_bytes(C.byref(c), 80, 20, 36, 5)
    assert 1 << 30 < need < 16 << 30, need             # a few GB of activations at B=80: trivial next to 288 GB
    offs = [L.vlt5_workspace_offset(C.byref(c), 80, 20, 36, 5, w) for w in range(7)]
    assert all(0 <= o < need for o in offs) and len(set(offs)) == 7
    # encoder-side offsets do not depend on T (greedy decoding re-plans with a growing T)
    a = L.vlt5_workspace_offset(C.byref(c), 80, 20, 36, 5, built.WS_ENC_EXT)
    b = L.vlt5_workspace_offset(C.byref(c), 80, 20, 36, 9, built.WS_ENC_EXT)
    assert a == b
    assert L.vlt5_workspace_bytes(C.byref(c), 0, 20, 36, 5) < 0


def test_gradient_release_plan_covers_every_bucket_once_and_follows_the_tuning(built):
    """VLT5.grad_release_plan (what the data-parallel wrapper cuts its collectives and -- sharded -- its chunk ownership by): every gradient
    bucket exactly once, the stacked cross-K/V bucket always in the decoder phase, the decoder layers' buckets in the encoder phase exactly when
    the engine lets their weight gradients ride in that phase's launches (vlt5_decoder_buckets_late: a function of configuration + tuning)."""
    from vqacl_amd import VLT5VQA, VLT5Config
    from vqacl_amd import _lib as L
    cfg = VLT5Config(d_model=64, d_kv=16, num_heads=4, d_ff=128, num_layers=4, num_decoder_layers=3, vocab_size=400, feat_dim=64)
    m = VLT5VQA(cfg, device=torch.device("cpu"))
    Ld, nb = 3, m._nbuckets
    assert nb == 3 + 1 + 4 + 1

    def covered(plan):
        seen = []
        for phase, lo, hi in plan:
            assert phase in (0, 1) and 0 <= lo <= hi <= nb
            seen += list(range(lo, hi))
        return seen
    plan = m.grad_release_plan()
    assert sorted(covered(plan)) == list(range(nb))
    assert (0, Ld, Ld + 1) in plan and (1, 0, Ld) in plan, plan            # default: shadowed, decoder layers released by the encoder phase
    assert plan[-1][2] == nb - 1 and plan[-2][1:] == (nb - 1, nb), "embeddings + norms before the lower half of the encoder, which comes last"
    for setting in (1, 3):                                                    # own launches / own launches under bucket events: the round-4 plan
        m.tuning = L.make_tuning(wgrad_shadow=setting)
        p2 = m.grad_release_plan()
        assert p2[0] == (0, 0, Ld + 1) and sorted(covered(p2)) == list(range(nb)), p2
    m.tuning = L.make_tuning()
    m.side_stream_enabled = True                                              # a side stream for the weight gradients: no shadowing either
    assert m.grad_release_plan()[0] == (0, 0, Ld + 1)
    # the plan's id (vlt5_step.release_plan_id: the backward phases refuse a plan that is not theirs) is a function of the triples
    m.side_stream_enabled = False
    ids = {}
    for setting in (0, 1):
        m.tuning = L.make_tuning(wgrad_shadow=setting)
        plan_, pid = m._release_plan()
        assert pid > 0 and plan_ == m.grad_release_plan()
        ids[setting] = pid
    assert ids[0] != ids[1]
    m.tuning = L.make_tuning(enc_cut=1)
    assert m._release_plan()[1] not in ids.values() and m._release_plan()[0] != plan
    m.tuning = L.make_tuning()
    assert m._release_plan() == (plan, ids[0])
    one = VLT5VQA(VLT5Config(d_model=64, d_kv=16, num_heads=4, d_ff=128, num_layers=1, num_decoder_layers=2, vocab_size=400, feat_dim=64),
                  device=torch.device("cpu"))
    assert sorted(covered(one.grad_release_plan())) == list(range(one._nbuckets)) and one.grad_release_plan()[0] == (0, 0, 3)


def test_moment_accumulator_of_the_dropout_comparison():
    """tests/trajectory_lib.py::_Moments (the statistics behind test_dropout_on_matches_the_oracle_in_distribution): running mean,
    summed unbiased variance and the two half sums against torch on stacked samples."""
    import trajectory_lib as T
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(7, 9, generator=g) * (1 + i % 3) + 0.5 for i in range(11)]
    m = T._Moments()
    for x in xs:
        m.add(a=x, loss=x.sum().reshape(1))
    st = torch.stack(xs).double().reshape(11, -1)
    assert torch.allclose(m.mean("a"), st.mean(0), atol=1e-12)
    assert abs(m.var_sum("a") - float(st.var(0, unbiased=True).sum())) < 1e-9
    assert torch.allclose(m.half[0]["a"], st[0::2].sum(0), atol=1e-12) and torch.allclose(m.half[1]["a"], st[1::2].sum(0), atol=1e-12)
    assert abs(m.var_sum("loss") - float(torch.stack([x.sum() for x in xs]).double().var(unbiased=True))) < 1e-9
    assert abs(T._cos(st[0], st[0] * 3.0) - 1.0) < 1e-12


def test_no_kernel_reuses_the_register_of_an_asynchronous_touch_request():
    """tools/check_touch_isa.py over the generated gfx950 ISA (hipcc -S, no GPU): a panel-touch request with a REGISTER destination
    (enc_attn.hip; the GEMM kernels' go to LDS since round 6) lands asynchronously -- no instruction may write that register before the
    next counted vmcnt wait.  Round 6 found exactly this as a memory fault after an unrelated edit moved the register allocation."""
    import subprocess
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_touch_isa.py")], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-1000:]
    assert "touch requests in" in res.stdout and ", 0 written before a counted wait" in res.stdout, res.stdout[-500:]
    n_req = int(res.stdout.strip().splitlines()[-1].split()[0])
    assert n_req >= 1, "the checker found no request at all: its pattern no longer matches the sources"
