#!/usr/bin/env python3
"""AdamW + gradient-norm kernels against the HBM roofline on a buffer of the model's size (226 M parameters, 30 + 4 B/param)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqacl_amd._lib import lib, ptr, stream_ptr
dev = torch.device("cuda")
n = 226_000_000
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 1e-3; m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
pb = torch.empty(n, device=dev, dtype=torch.bfloat16)
tot = torch.ones(1, device=dev); part = torch.empty(lib().vlt5_sqnorm_blocks(n), device=dev)
def step(t):
    lib().vlt5_sqnorm(ptr(g), n, ptr(part), ptr(tot), 0, stream_ptr())
    lib().vlt5_adamw_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(pb), n, 1e-4, 0.9, 0.999, 1e-6, 0.01, t, ptr(tot), 5.0, 1, stream_ptr())
for t in range(1, 4):
    step(t)
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
e[0].record()
for t in range(4, 14):
    lib().vlt5_sqnorm(ptr(g), n, ptr(part), ptr(tot), 0, stream_ptr())
e[1].record()
for t in range(4, 14):
    lib().vlt5_adamw_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(pb), n, 1e-4, 0.9, 0.999, 1e-6, 0.01, t, ptr(tot), 5.0, 1, stream_ptr())
e[2].record()
torch.cuda.synchronize()
sq, ad = e[0].elapsed_time(e[1]) / 10, e[1].elapsed_time(e[2]) / 10
print(f"sqnorm {sq * 1e3:7.1f} us  {4 * n / sq / 1e9:6.2f} TB/s   adamw {ad * 1e3:7.1f} us  {30 * n / ad / 1e9:6.2f} TB/s")
