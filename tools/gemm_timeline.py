#!/usr/bin/env python3
"""Per-workgroup timeline of one GEMM launch (needs the -DGEMM_TIMELINE debug library: VLT5_LIB=.../libvlt5_tl.so).
usage: VLT5_LIB=$PWD/vqacl_amd/libvlt5_tl.so python tools/gemm_timeline.py M N K akm bkm tile_m tile_n [f32]
Points (shader-clock ticks, wave 0 of each workgroup): t0 kernel entry, t1 prologue loads issued, t2 first k-tile landed and
barrier passed, t3 main loop done, t4 epilogue stores issued, t5 stores acknowledged."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqacl_amd import _lib, ops  # noqa: E402

M, N, K, akm, bkm, tm, tn = (int(x) for x in sys.argv[1:8])
f32 = len(sys.argv) > 8 and sys.argv[8] == "f32"
dev = torch.device("cuda")
A = torch.randn((K, M) if akm else (M, K), device=dev).to(torch.bfloat16)
B = torch.randn((K, N) if bkm else (N, K), device=dev).to(torch.bfloat16)
out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
ntiles = -(-M // tm) * -(-N // tn)
buf = torch.zeros(ntiles * 8, dtype=torch.int64, device=dev)
lib = C.CDLL(_lib.LIB_PATH)
lib.vlt5dbg_set_timeline.argtypes = [C.c_void_p]
run = lambda: ops.gemm(A, B, M, N, K, a_kmajor=bool(akm), b_kmajor=bool(bkm), out=out, tile=(tm, tn))
for _ in range(3):
    run()
torch.cuda.synchronize()
assert lib.vlt5dbg_set_timeline(C.c_void_p(buf.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
t = buf.cpu().numpy().astype(np.uint64).reshape(ntiles, 8)
hw = t[:, 6]
xcc = (hw & np.uint64(0xF)).astype(int)
hwid = (hw >> np.uint64(32)).astype(np.int64)
cu = ((hwid >> 8) & 0xF).astype(int)
se = ((hwid >> 13) & 0x7).astype(int)
t = t[:, :6].astype(np.int64)
# the counter is per XCD (not synchronised across XCDs): offsets are taken against the first workgroup of the same XCD
base = np.zeros(ntiles, dtype=np.int64)
for x in range(16):
    sel = xcc == x
    if sel.any():
        base[sel] = t[sel, 0].min()
span = (t[:, 5] - base).max()
nk = K // 64
tick_us = float(os.environ.get("TICKS_PER_US", "100"))      # s_memtime ticks at the 100 MHz reference clock on gfx950
print(f"M={M} N={N} K={K} akm={akm} bkm={bkm} tile={tm}x{tn}: {ntiles} workgroups, event time {us:.1f} us, tick span {span} "
      f"= {span / tick_us:.1f} us at {tick_us:.0f} ticks/us")


def stat(name, v):
    v = v / tick_us
    print(f"  {name:34s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f} us")


stat("start offset (t0 - first t0)", t[:, 0] - base)
stat("setup + prologue issue (t1-t0)", t[:, 1] - t[:, 0])
stat("first k-tile landed (t2-t1)", t[:, 2] - t[:, 1])
stat(f"main loop (t3-t2), {nk} k-steps", t[:, 3] - t[:, 2])
stat("  per k-step", (t[:, 3] - t[:, 2]) / max(nk, 1))
stat("epilogue issue (t4-t3)", t[:, 4] - t[:, 3])
stat("store drain (t5-t4)", t[:, 5] - t[:, 4])
stat("workgroup life (t5-t0)", t[:, 5] - t[:, 0])
stat("end offset (t5 - first t0 of xcd)", t[:, 5] - base)
uniq = len(set(zip(xcc.tolist(), se.tolist(), cu.tolist())))
print(f"  distinct (xcc, se, cu) = {uniq}; workgroups per xcc = {np.bincount(xcc, minlength=8).tolist()}")
