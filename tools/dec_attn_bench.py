#!/usr/bin/env python3
"""Fused decoder attention sublayer kernels against the three launches they replace (projection GEMM, core, output projection as the
engine issues it: split-K slabs), back to back on the base model's decoder shapes (B = 80, T = 5, Sx = 58).
usage: python tools/dec_attn_bench.py [B T]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vqacl_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 80
T = int(sys.argv[2]) if len(sys.argv) > 2 else 5
Tk, H, d = 58, 12, 768
inner = H * 64
dev = torch.device("cuda")
BF = torch.bfloat16
g = torch.Generator().manual_seed(1)
xn = torch.randn(B * T, d, generator=g).to(BF).to(dev)
wqkv = (torch.randn(3 * inner, d, generator=g) * d ** -0.5).to(BF).to(dev)
wq = (torch.randn(inner, d, generator=g) * d ** -0.5).to(BF).to(dev)
wo = (torch.randn(d, inner, generator=g) * inner ** -0.5).to(BF).to(dev)
bias = torch.randn(H, T, T, generator=g).to(dev)
kv = torch.randn(B, Tk, 12 * 2 * inner, generator=g).to(BF).to(dev)
k, v = kv[:, :, 2 * inner:3 * inner], kv[:, :, 3 * inner:4 * inner]
km = torch.ones(B, Tk, device=dev)


def timeit(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000


def unfused_self():
    qkv = ops.gemm(xn, wqkv, B * T, 3 * inner, d).view(B, T, 3 * inner)
    ctx, _ = ops.attn_fwd(qkv[:, :, :inner], qkv[:, :, inner:2 * inner], qkv[:, :, 2 * inner:], H, 64, bias=bias, causal=True, drop_p=0.1, drop_seed=7)
    return ops.gemm(ctx.view(B * T, inner), wo, B * T, d, inner, out_f32=True)


def unfused_cross():
    q = ops.gemm(xn, wq, B * T, inner, d).view(B, T, inner)
    ctx, _ = ops.attn_fwd(q, k, v, H, 64, key_mask=km, mask_value=-1e9, drop_p=0.1, drop_seed=7)
    return ops.gemm(ctx.view(B * T, inner), wo, B * T, d, inner, out_f32=True)


print(f"B={B} T={T}: host-bound loops include the Python wrapper's allocations; compare like with like")
print(f"self : three launches {timeit(unfused_self):7.2f} us   fused {timeit(lambda: ops.dec_attn_fused(xn, wqkv, wo, B, T, H, bias=bias, drop_p=0.1, drop_seed=7)):7.2f} us")
print(f"cross: three launches {timeit(unfused_cross):7.2f} us   fused {timeit(lambda: ops.dec_attn_fused(xn, wq, wo, B, T, H, k=k, v=v, key_mask=km, mask_value=-1e9, drop_p=0.1, drop_seed=7)):7.2f} us")

# per-phase shader clocks of wave 0 (instrumented build), median over workgroups
import ctypes as C  # noqa: E402
from vqacl_amd._lib import lib  # noqa: E402
L = lib()
L.vlt5dbg_dec_attn_timeline.argtypes = [C.c_void_p]
names = ["setup", "prologue req", "first tile", "k-loop", "Wo req", "hand-over+stores", "core", "Wo landed", "barrier", "phase3+stores", "drain"]
for label, fn in (("self", lambda: ops.dec_attn_fused(xn, wqkv, wo, B, T, H, bias=bias, drop_p=0.1, drop_seed=7)),
                  ("cross", lambda: ops.dec_attn_fused(xn, wq, wo, B, T, H, k=k, v=v, key_mask=km, mask_value=-1e9, drop_p=0.1, drop_seed=7))):
    nwg = ((B + 5) // 6) * H  # (T = 5: six samples per workgroup)
    buf = torch.zeros(nwg * 12, dtype=torch.int64, device=dev)
    L.vlt5dbg_dec_attn_timeline(C.c_void_p(buf.data_ptr()))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    L.vlt5dbg_dec_attn_timeline(C.c_void_p(0))
    t = buf.view(nwg, 12).cpu().double()
    d = (t[:, 1:] - t[:, :-1]).median(0).values
    tot = (t[:, 11] - t[:, 0]).median()
    print(label, "cycles:", "  ".join(f"{n} {int(x)}" for n, x in zip(names, d.tolist())), f"| total {int(tot)}")
