#!/usr/bin/env python3
"""Greedy decoding with the token-step replayed from a HIP graph against enqueueing its ~100 launches every step: same process, alternating."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_cpu import Cfg, synthetic_batch  # noqa: E402  (the synthetic-input recipe only)
from vqacl_amd import VLT5Config, VLT5VQA  # noqa: E402

B = 80
dev = torch.device("cuda")
torch.manual_seed(1)
model = VLT5VQA(VLT5Config(dropout_rate=0.1), device=dev)
model.eval()
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic_batch(Cfg(), B=B, L=20, V=36, T=5, seed=3, task_id=0).items()}
fb = (batch["vis_feats"], batch["boxes"])


def timed(n, reps=5):
    for _ in range(2):
        model.greedy_generate(batch["input_ids"], fb, max_length=n, eos_token_id=-1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model.greedy_generate(batch["input_ids"], fb, max_length=n, eos_token_id=-1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for rnd in range(3):
    for graph in (False, True):
        model.decode_graph = graph
        t3, t20 = timed(3), timed(20)
        print(f"round {rnd + 1}  graph={int(graph)}  max_length 3: {t3:6.2f} ms   max_length 20: {t20:6.2f} ms   -> {(t20 - t3) / 17:6.4f} ms per token-step", flush=True)
