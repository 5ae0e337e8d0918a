#!/usr/bin/env python3
"""EXPERIMENT (needs tools/experiments/stream_placement.patch applied; measured in round 6 and not kept: profiles/r06_i_*).
clip + AdamW of the real optimizer (nine launches over the model's runs) with the streams placed (vqacl_amd/placement.py) and as
separate allocations, in one process on one box: the optimizer step alone (events around opt.step() after a real backward) and the
whole train step.      python tools/placement_ab.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch  # noqa: E402
from vqacl_amd import VLT5VQA, VLT5Config, FusedAdamW, reference_param_groups  # noqa: E402

dev = torch.device("cuda")
batch = {k: v.to(dev) for k, v in synthetic_batch(80, seed=1).items()}
for rnd in range(2):
    for mode in ("0", "1"):
        os.environ["VQACL_PLACEMENT"] = mode
        torch.manual_seed(1)
        model = VLT5VQA(VLT5Config(dropout_rate=0.1), device=dev)
        model.train()
        opt = FusedAdamW(reference_param_groups(model, 0.01), model, lr=1e-4, eps=1e-6, max_grad_norm=5.0)

        def fwd_bwd():
            res = model.train_step(batch, 0, 0.5, 0.3)
            res["loss"].backward()

        def clear():
            for p in model.parameters():
                p.grad = None
        for _ in range(4):
            fwd_bwd(); opt.step(); clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fwd_bwd(); opt.step(); clear()
        torch.cuda.synchronize()
        step_ms = (time.perf_counter() - t0) / 20 * 1e3
        adam = 0.0
        for _ in range(10):
            fwd_bwd()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); opt.step(); e1.record(); e1.synchronize()
            adam += e0.elapsed_time(e1)
            clear()
        info = model.placement_info
        t = info.get("adamw_us_by_candidate") or {}
        print(f"round {rnd}  VQACL_PLACEMENT={mode}: step {step_ms:6.3f} ms   clip + AdamW {adam / 10:6.3f} ms   {info.get('placement')}"
              + (f"  (trials {min(t.values())} ... {max(t.values())} us; separate sets {[v for k, v in t.items() if k.startswith('separate')]})" if t else ""), flush=True)
        del model, opt
        torch.cuda.empty_cache()
