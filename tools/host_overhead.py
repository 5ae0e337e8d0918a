#!/usr/bin/env python3
"""How long does the HOST need to enqueue one train step (no synchronisation)?  If this is well below the GPU step time the
path is GPU-bound and graph capture would not help."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_cpu import synthetic_batch, Cfg
from vqacl_amd import VLT5VQA, VLT5Config, FusedAdamW, reference_param_groups

dev = torch.device("cuda")
model = VLT5VQA(VLT5Config(dropout_rate=0.1), device=dev)
model.train()
opt = FusedAdamW(reference_param_groups(model, 0.01), model, lr=1e-4, eps=1e-6, max_grad_norm=5.0)
batch = {k: v.to(dev) for k, v in synthetic_batch(Cfg(), B=80, L=20, V=36, T=5, seed=1).items()}

def step():
    res = model.train_step(batch, 0, 0.5, 0.3)
    res["loss"].backward()
    opt.step()
    for p in model.parameters():
        p.grad = None

for _ in range(5):
    step()
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for _ in range(20):
    t0 = time.perf_counter()
    step()
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / 20
print(f"host enqueue per step: median {sorted(host)[10]*1e3:.2f} ms, min {min(host)*1e3:.2f} ms; wall per step {wall*1e3:.2f} ms")
torch.cuda.synchronize()
t0 = time.perf_counter(); step(); th = time.perf_counter() - t0; torch.cuda.synchronize(); tw = time.perf_counter() - t0
print(f"single step from idle: host {th*1e3:.2f} ms, until GPU done {tw*1e3:.2f} ms")
