#!/usr/bin/env python3
"""Phase timeline of the fused q|k|v projection + attention kernel (csrc/enc_attn.hip, instrumented build) at the benched shape, and
its launch duration against the two launches it replaces.  usage (GPU box): python tools/enc_attn_timeline.py [B S H d]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqacl_amd import ops  # noqa: E402
from vqacl_amd._lib import lib  # noqa: E402

B, S, H, d = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (80, 56, 12, 768)
dev = torch.device("cuda")
BF = torch.bfloat16
g = torch.Generator().manual_seed(0)
inner = H * 64
xn = torch.randn(B * S, d, generator=g).to(BF).to(dev)
w = (torch.randn(3 * inner, d, generator=g) * d ** -0.5).to(BF).to(dev)
bias = torch.randn(H, 20, 20, generator=g).to(dev)
km = torch.ones(B, S, device=dev)
kw = dict(bias=bias, key_mask=km, drop_p=0.1, drop_seed=5)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def unfused():
    qkv = ops.gemm(xn, w, B * S, 3 * inner, d).view(B, S, 3 * inner)
    ops.attn_fwd(qkv[:, :, :inner], qkv[:, :, inner:2 * inner], qkv[:, :, 2 * inner:], H, 64, **kw)


print(f"B={B} S={S} H={H} d={d}: fused {timed(lambda: ops.qkv_attn_fwd(xn, w, B, S, H, **kw)):.1f} us, "
      f"gemm + attention core {timed(unfused):.1f} us (python-side launches included)")
L = C.CDLL(lib()._name)
nwg = ((B + 1) // 2) * (H // 2)
buf = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
L.vlt5dbg_qkv_attn_timeline.argtypes = [C.c_void_p]
L.vlt5dbg_qkv_attn_timeline(C.c_void_p(buf.data_ptr()))
for _ in range(3):
    ops.qkv_attn_fwd(xn, w, B, S, H, **kw)
torch.cuda.synchronize()
L.vlt5dbg_qkv_attn_timeline(C.c_void_p(0))
t = buf.view(nwg, 16).cpu().double()
# stamp order in the kernel: 0 start, 1 prologue issued, 2 first k-tile landed, 3 main loop done, 8 addends requested, 9 barrier 1,
# 10 tiles written, 4 barrier 2, 5 q|k|v stores issued, 6 core done, 7 stores drained
order = [0, 1, 2, 3, 8, 9, 10, 4, 5, 11, 12, 13, 6, 7]
names = ["prologue issue", "first k-tile landed", "k-steps 2..n", "addend request", "barrier 1 (incl. load wait)", "tile writes",
         "barrier 2", "q|k|v store issue", "core: addend finish", "core: K/Q reads + score MFMAs", "core: softmax + dropout + pack",
         "core: V^T reads + P.V + ctx stores", "store drain"]
print(f"{'phase':34s} {'mean clk':>9s} {'min':>8s} {'max':>8s}")
for i, nm in enumerate(names):
    dlt = t[:, order[i + 1]] - t[:, order[i]]
    print(f"{nm:34s} {dlt.mean():9.0f} {dlt.min():8.0f} {dlt.max():8.0f}")
tot = t[:, 7] - t[:, 0]
print(f"{'workgroup total':34s} {tot.mean():9.0f} {tot.min():8.0f} {tot.max():8.0f}")
