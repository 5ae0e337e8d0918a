#!/usr/bin/env python3
"""Is the greedy-decoding loop bound by the host's launch rate?  Behind one encoder pass: 19 vlt5_decoder_step_greedy calls enqueued without any
host synchronisation; prints the host time to enqueue them and the time until the GPU has finished them."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_cpu import Cfg, synthetic_batch  # noqa: E402  (the synthetic-input recipe only)
from vqacl_amd import VLT5Config, VLT5VQA  # noqa: E402
from vqacl_amd import _lib as L  # noqa: E402
from vqacl_amd._lib import check, lib, ptr, stream_ptr  # noqa: E402

B = 80
dev = torch.device("cuda")
torch.manual_seed(1)
model = VLT5VQA(VLT5Config(dropout_rate=0.1), device=dev)
model.eval()
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic_batch(Cfg(), B=B, L=20, V=36, T=5, seed=3, task_id=0).items()}
fb = (batch["vis_feats"], batch["boxes"])
Tcap = 20
out = model.greedy_generate(batch["input_ids"], fb, max_length=Tcap, eos_token_id=-1)       # sizes the arena, fills the cross-K/V of this batch
cfg = model.cfg
feats, boxes, V, ref = model._visual_inputs(fb)
dims = (B, 20, V, Tcap)
st = dict(dims=dims, training=False, seed=0, feats=feats, boxes=boxes, feat_ref=ref, input_ids=batch["input_ids"],
          labels=torch.zeros(B, Tcap, dtype=torch.long, device=dev), enc_lut=model._lut(20, 20, True), dec_lut=model._lut(Tcap, Tcap, False))
c = cfg.c_struct()
cs = model._make_step(st)
inner = cfg.num_heads * cfg.d_kv
cache = torch.empty(cfg.num_decoder_layers, B, Tcap, 2 * inner, device=dev, dtype=torch.bfloat16)
cur = torch.full((B,), cfg.decoder_start_token_id, dtype=torch.long, device=dev)
toks = torch.zeros(B, Tcap, dtype=torch.long, device=dev)
done = torch.zeros(B, dtype=torch.int32, device=dev)
g = L.GreedyDesc()
g.tokens, g.kv_cache, g.out_tokens, g.out_ld, g.done = ptr(cur), ptr(cache), ptr(toks), Tcap, ptr(done)
g.eos_id, g.pad_id = -1, cfg.pad_token_id
stream = stream_ptr()
for rep in range(4):
    done.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(Tcap - 1):
        g.t = t
        check(lib().vlt5_decoder_step_greedy(C.byref(c), C.byref(cs), C.byref(g), stream), "step")
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"19 token-steps: host enqueue {(t1 - t0) * 1e3:6.2f} ms ({(t1 - t0) / 19 * 1e3:.3f} per step), until the GPU is done {(t2 - t0) * 1e3:6.2f} ms "
          f"({(t2 - t0) / 19 * 1e3:.3f} per step)", flush=True)
