#!/usr/bin/env python3
"""Embedding-gradient scatter (deterministic) at the step's two shapes: B*L = 1600 text rows with ~35 % pad, B*T = 400 decoder rows."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_sweep import timed_graph
from vqacl_amd._lib import lib, ptr, stream_ptr
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
for B, T in ((80, 20), (80, 5)):
    d, vocab = 768, 32200
    ids = torch.randint(2, 32000, (B, T), generator=g)
    lens = torch.randint(max(1, T // 3), T + 1, (B,), generator=g)
    ids = (ids * (torch.arange(T)[None] < lens[:, None])).to(dev)
    dout = torch.randn(B, T, d, device=dev)
    tab = torch.zeros(vocab, d, device=dev)
    sc = torch.empty(lib().vlt5_embed_bwd_scratch_bytes(B, T, d), dtype=torch.uint8, device=dev)
    fn = lambda: lib().vlt5_embed_bwd(ptr(ids), ptr(dout), T * d, d, ptr(tab), B, T, d, vocab, 0.1, 7, T, 0, ptr(sc), stream_ptr())
    print(f"B*T = {B * T}: {timed_graph(fn):.2f} us per launch ({int((ids == 0).sum())} pad rows)")
