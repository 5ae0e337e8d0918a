#!/usr/bin/env python3
"""Phase timeline of attn_bwd_kernel (needs a -DATTN_TIMELINE build: bash tools/build_variant.sh atl -DATTN_TIMELINE;
VLT5_LIB=$PWD/vqacl_amd/libvlt5_atl.so python tools/attn_bwd_timeline.py [enc|cross|self])."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqacl_amd import _lib, ops  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "enc"
B, H, dk = 80, 12, 64
Tq, Tk = {"enc": (56, 56), "cross": (5, 58), "self": (5, 5)}[kind]
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
mk = lambda T: (torch.randn(B, T, H * dk, generator=g) * 0.5).to(torch.bfloat16).to(dev)
q, k, v, do = mk(Tq), mk(Tk), mk(Tk), mk(Tq)
bias = torch.randn(H, Tq, Tk, generator=g).to(dev) if kind != "cross" else None
mask = torch.ones(B, Tk, device=dev) if kind != "self" else None
kw = dict(bias=bias, key_mask=mask, mask_value=-1e9, causal=(kind == "self"), drop_p=0.1, drop_seed=7)
ctx, lse = ops.attn_fwd(q, k, v, H, dk, **kw)
run = lambda: ops.attn_bwd(q, k, v, do, lse, H, dk, want_dbias=bias is not None, **kw)
for _ in range(3):
    run()
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = torch.zeros(B * H * 8, dtype=torch.int64, device=dev)
lib.vlt5dbg_set_attn_timeline.argtypes = [C.c_void_p]
assert lib.vlt5dbg_set_attn_timeline(C.c_void_p(buf.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(B * H, 8).astype(np.int64)
print(f"attn_bwd {kind}: B={B} H={H} Tq={Tq} Tk={Tk} dk={dk}, {B * H} workgroups; launch {e0.elapsed_time(e1) * 1e3:.1f} us (event pair, incl. the python call)")
names = ["addends + tile loads issued", "tiles landed + stored to LDS + barrier", "phase A: scores, dP, softmax bwd, dbias, dQ", "Pd / dS hand-over (2 barriers)",
         "phase B: dV, dK", "stores landed"]
print(f"{'phase':50s} {'mean clk':>9s} {'p10':>8s} {'p90':>8s}")
for i, n in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print(f"{n:50s} {d.mean():9.0f} {np.percentile(d, 10):8.0f} {np.percentile(d, 90):8.0f}")
life = t[:, 6] - t[:, 0]
print(f"{'workgroup total':50s} {life.mean():9.0f} {np.percentile(life, 10):8.0f} {np.percentile(life, 90):8.0f}")
print(f"first start -> last end: {(t[:, 6].max() - t[:, 0].min())} shader clocks (the clocks of different CUs are not synchronised: indicative only)")
