#!/usr/bin/env python3
"""Greedy-decoding throughput of test_step: key/value-cached incremental decoder against re-decoding the
prefix with the training kernels.  VL-T5-base, B=80 (--batch), 36 regions, 20 question tokens, max_length 20, random weights
(EOS is practically never produced, so every row decodes the full 19 steps)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_cpu import Cfg, synthetic_batch  # noqa: E402  (the synthetic-input recipe only)
from vqacl_amd import VLT5Config, VLT5VQA  # noqa: E402

B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 80
dev = torch.device("cuda")
torch.manual_seed(1)
LARGE = "--large" in sys.argv                       # VL-T5-large (BASELINE configs[4]): d = 1024, 16 heads, d_ff = 4096, 24 + 24 layers
kw = dict(d_model=1024, num_heads=16, d_ff=4096, num_layers=24) if LARGE else {}
model = VLT5VQA(VLT5Config(dropout_rate=0.1, **kw), device=dev)
model.eval()
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic_batch(Cfg(), B=B, L=20, V=36, T=5, seed=3, task_id=0).items()}
fb = (batch["vis_feats"], batch["boxes"])
def run(name, fast, **kw):
    model.tuning.decode_fast = 2 if fast else 1
    for _ in range(2):
        out = model.greedy_generate(batch["input_ids"], fb, max_length=20, eos_token_id=-1, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        out = model.greedy_generate(batch["input_ids"], fb, max_length=20, eos_token_id=-1, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    steps = out.shape[1] - 1
    print(f"{name:28s} B={B} generated {steps} tokens/row: {dt * 1e3:7.2f} ms per batch (encoder + cross-K/V included), "
          f"{dt / steps * 1e3:6.3f} ms per step, {B * steps / dt:9.0f} tokens/s, {B / dt:7.0f} samples/s", flush=True)
    return out


FAST_ONLY = "--fast-only" in sys.argv
a = run("kv-cache, decode kernels", True, use_cache=True)
if not FAST_ONLY:
    b = run("kv-cache, tiled launches", False, use_cache=True)
    run("recompute the prefix", False, use_cache=False)
    print(f"tokens equal between the two cached paths: {int((a == b).sum())} of {a.numel()} (random weights: small logit margins)")
if True:
    # the token steps alone: time 19 vlt5_decoder_step_greedy calls behind one encoder pass with HIP events (no host sync inside)
    import ctypes as C
    from vqacl_amd import _lib as L
    from vqacl_amd._lib import check, lib, ptr, stream_ptr
    for fast in ((True,) if FAST_ONLY else (True, False)):
        model.tuning.decode_fast = 2 if fast else 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # greedy_generate does encoder + steps; the encoder part alone:
        ids = batch["input_ids"]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            model.greedy_generate(ids, fb, max_length=2, eos_token_id=-1)
        torch.cuda.synchronize()
        t_enc = (time.perf_counter() - t0) / 5
        t0 = time.perf_counter()
        for _ in range(5):
            model.greedy_generate(ids, fb, max_length=20, eos_token_id=-1)
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / 5
        print(f"{'decode kernels' if fast else 'tiled launches':16s}: encoder + first step {t_enc * 1e3:6.2f} ms; 18 further steps {(t_all - t_enc) * 1e3:6.2f} ms "
              f"= {(t_all - t_enc) / 18 * 1e3:6.3f} ms per token-step", flush=True)
