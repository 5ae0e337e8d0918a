#!/usr/bin/env python3
"""Launch one GEMM shape repeatedly (for rocprofv3 PMC passes): python tools/gemm_one.py M N K akm bkm tile_m tile_n [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqacl_amd import ops  # noqa: E402

M, N, K, akm, bkm, tm, tn = (int(x) for x in sys.argv[1:8])
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 10
dev = torch.device("cuda")
A = torch.randn((K, M) if akm else (M, K), device=dev).to(torch.bfloat16)
B = torch.randn((K, N) if bkm else (N, K), device=dev).to(torch.bfloat16)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(reps):
    ops.gemm(A, B, M, N, K, a_kmajor=bool(akm), b_kmajor=bool(bkm), out=out, tile=(tm, tn))
torch.cuda.synchronize()
