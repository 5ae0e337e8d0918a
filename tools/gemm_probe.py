#!/usr/bin/env python3
"""Correctness + HIP-event timing of ONE tile choice over a list of shapes (experiments on kernel variants).
usage: python tools/gemm_probe.py TM TN [shape-set]     shape-set: wide (N >= 2304) | narrow (N = 768) | wgrad"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqacl_amd import ops  # noqa: E402

dev = torch.device("cuda")
BF = torch.bfloat16
tm, tn = int(sys.argv[1]), int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "narrow"
SETS = {
    "wide": [(4480, 3072, 768, 0, 0, 0), (4480, 2304, 768, 0, 0, 0), (4480, 3072, 768, 0, 1, 0), (4640, 18432, 768, 0, 0, 0), (400, 32200, 768, 0, 0, 1)],
    "narrow": [(4480, 768, 3072, 0, 0, 1), (4480, 768, 3072, 0, 1, 1), (4480, 768, 768, 0, 0, 1), (4480, 768, 768, 0, 1, 0), (4480, 768, 2304, 0, 1, 1),
               (4480, 2304, 768, 0, 0, 0), (1000, 520, 200, 0, 0, 1)],
    "wgrad": [(3072, 768, 4480, 1, 1, 1), (768, 3072, 4480, 1, 1, 1), (2304, 768, 4480, 1, 1, 1), (768, 768, 4480, 1, 1, 1)],
}
print("tile", tm, tn, {k: v for k, v in os.environ.items() if k.startswith("VLT5_GEMM")})
for M, N, K, akm, bkm, f32 in SETS[which]:
    A = torch.randn((K, M) if akm else (M, K), device=dev).to(BF)
    B = torch.randn((K, N) if bkm else (N, K), device=dev).to(BF)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else BF)
    ref = ((A.float().t() if akm else A.float()) @ (B.float() if bkm else B.float().t()))
    kw = dict(a_kmajor=bool(akm), b_kmajor=bool(bkm), out=out, tile=(tm, tn))
    ops.gemm(A, B, M, N, K, **kw)
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    for _ in range(3):
        ops.gemm(A, B, M, N, K, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm(A, B, M, N, K, **kw)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"M={M:5d} N={N:5d} K={K:5d} akm={akm} bkm={bkm} f32={f32}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s  rel err {err:.2e}")
