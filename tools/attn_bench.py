#!/usr/bin/env python3
"""HIP-event timing of the attention kernels at the model's shapes (launch-only loop).  usage: python tools/attn_bench.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqacl_amd import ops  # noqa: E402
from vqacl_amd._lib import lib, stream_ptr  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
print({k: v for k, v in os.environ.items() if k.startswith("VLT5_ATTN")})
for name, B, H, Tq, Tk, causal, masked in (("encoder self", 80, 12, 56, 56, False, True), ("decoder self", 80, 12, 5, 5, True, False),
                                           ("decoder cross", 80, 12, 5, 58, False, True)):
    inner = H * 64
    q = torch.randn(B, Tq, inner, device=dev).to(BF)
    k = torch.randn(B, Tk, inner, device=dev).to(BF)
    v = torch.randn(B, Tk, inner, device=dev).to(BF)
    do = torch.randn(B, Tq, inner, device=dev).to(BF)
    bias = torch.randn(H, min(Tq, 20), min(Tk, 20), device=dev) if not masked or Tq == Tk else None
    km = torch.ones(B, Tk, device=dev) if masked else None
    kw = dict(bias=bias, key_mask=km, mask_value=-10000.0, causal=causal, drop_p=0.1, drop_seed=5)
    ctx, lse = ops.attn_fwd(q, k, v, H, 64, **kw)

    def t(fn, n=100):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    tf = t(lambda: ops.attn_fwd(q, k, v, H, 64, **kw))
    tb = t(lambda: ops.attn_bwd(q, k, v, do, lse, H, 64, want_dbias=bias is not None, **kw))
    print(f"{name:14s} B={B} H={H} Tq={Tq} Tk={Tk}: fwd {tf:6.2f} us   bwd {tb:6.2f} us   (Python-side launch path included)")
