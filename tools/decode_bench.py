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
model = VLT5VQA(VLT5Config(dropout_rate=0.1), device=dev)
model.eval()
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic_batch(Cfg(), B=B, L=20, V=36, T=5, seed=3, task_id=0).items()}
fb = (batch["vis_feats"], batch["boxes"])
for name, kw in (("kv-cache", dict(use_cache=True)), ("recompute", dict(use_cache=False))):
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
    print(f"{name:10s} B={B} generated {steps} tokens/row: {dt * 1e3:7.2f} ms per batch, {dt / steps * 1e3:6.3f} ms per step, "
          f"{B * steps / dt:9.0f} tokens/s, {B / dt:7.0f} samples/s")
