#!/usr/bin/env python3
"""Graph-replayed timing of chosen GEMM launches: python tools/gemm_probe2.py "M N K akm bkm tm tn [batch]" ...  (f32 output)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_sweep import timed_graph
from vqacl_amd import ops
dev = torch.device("cuda")
for spec in sys.argv[1:]:
    v = [int(x) for x in spec.split()]
    M, N, K, akm, bkm, tm, tn = v[:7]
    batch = v[7] if len(v) > 7 else 1
    sk = v[8] if len(v) > 8 else 1
    A = torch.randn((batch, K, M) if akm else (batch, M, K), device=dev).to(torch.bfloat16)
    B = torch.randn((batch, K, N) if bkm else (batch, N, K), device=dev).to(torch.bfloat16)
    out = torch.empty(batch, M, N, device=dev, dtype=torch.float32)
    fn = lambda: ops.gemm(A[0], B[0], M, N, K, a_kmajor=bool(akm), b_kmajor=bool(bkm), out=out[0], tile=(tm, tn), batch=batch, split_k=sk,
                          batch_strides=(A.stride(0), B.stride(0), out.stride(0)))
    us = timed_graph(fn)
    print(f"M={M} N={N} K={K} akm={akm} bkm={bkm} tile={tm}x{tn} batch={batch} sk={sk}: {us:8.1f} us  {2.0 * batch * M * N * K / us / 1e6:7.1f} TF", flush=True)
