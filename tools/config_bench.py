#!/usr/bin/env python3
"""Train-step throughput of the other BASELINE configurations (side data; bench.py measures configs[1]):
python tools/config_bench.py [steps]   -> c4 NExT-QA shapes (V=16 / V=32, L=23, T=6, B=80), c5 VL-T5-large (B=32), base B=4 (c1 shape)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_cpu import Cfg, synthetic_batch          # synthetic-input recipe only  # noqa: E402
from vqacl_amd import FusedAdamW, VLT5Config, VLT5VQA, reference_param_groups  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
dev = torch.device("cuda:0")
CASES = [("c1 base B=4", {}, 4, 20, 36, 5), ("c2 base B=80", {}, 80, 20, 36, 5), ("c4 nextqa V=16", dict(n_ques=8), 80, 23, 16, 6),
         ("c4 nextqa V=32", dict(n_ques=8), 80, 23, 32, 6),
         ("c5 large B=32", dict(d_model=1024, num_heads=16, d_ff=4096, num_layers=24), 32, 20, 36, 5)]
for name, kw, B, L, V, T in CASES:
    cfg = VLT5Config(dropout_rate=0.1, **kw)
    torch.manual_seed(1)
    model = VLT5VQA(cfg, device=dev)
    model.train()
    opt = FusedAdamW(reference_param_groups(model, 0.01), model, lr=1e-4, eps=1e-6, max_grad_norm=5.0)
    ocfg = Cfg(d_model=cfg.d_model, num_heads=cfg.num_heads, d_ff=cfg.d_ff, num_layers=cfg.num_layers, n_ques=cfg.n_ques)
    batch = {k: v.to(dev) for k, v in synthetic_batch(ocfg, B=B, L=L, V=V, T=T, seed=3).items()}

    def step():
        model.train_step(batch, 0, 0.5, 0.3)["loss"].backward()
        opt.step()
        for p in model.parameters():
            p.grad = None
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{name:18s} {B * 1e3 / ms:9.1f} samples/s  {ms:7.2f} ms/step")
    del model, opt, batch
    torch.cuda.empty_cache()
