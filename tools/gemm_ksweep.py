#!/usr/bin/env python3
"""Fixed cost vs per-k-step cost of the GEMM kernel: time M x N outputs for growing K (HIP events)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqacl_amd import ops
from tools.gemm_sweep import timed

dev = torch.device("cuda")
BF = torch.bfloat16
for (M, N, of32, bkm) in ((4480, 768, True, False), (4480, 768, False, False), (4480, 3072, False, False), (4480, 2304, False, False), (4480, 768, True, True), (400, 768, True, False), (400, 2304, False, False)):
    line = []
    for K in (64, 128, 256, 512, 768, 1536, 3072):
        A = torch.randn(M, K, device=dev).to(BF)
        B = torch.randn((K, N) if bkm else (N, K), device=dev).to(BF)
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if of32 else BF)
        us = timed(lambda: ops.gemm(A, B, M, N, K, b_kmajor=bkm, out=out), reps=20)
        line.append(f"K={K}:{us:6.1f}us")
    print(f"M={M} N={N} out={'f32' if of32 else 'bf16'} bkm={int(bkm)} | " + "  ".join(line), flush=True)
