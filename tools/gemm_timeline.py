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
def analyse(t, label):
    rt0, rt1 = t[:, 7].astype(np.int64), t[:, 6].astype(np.int64)      # 100 MHz device-wide clock: start / end of each workgroup
    t = t[:, :6].astype(np.int64)
    nk = K // 64
    life = t[:, 5] - t[:, 0]
    tick_us = float(np.median(life / np.maximum(rt1 - rt0, 1))) * 100.0   # shader ticks per microsecond
    print(f"{label}: shader clock ~ {tick_us:.0f} ticks/us")

    def stat(name, v, per=tick_us):
        v = v / per
        print(f"  {name:36s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f} us")

    stat("start offset (first workgroup = 0)", rt0 - rt0.min(), 100.0)
    stat("end offset", rt1 - rt0.min(), 100.0)
    stat("setup + prologue issue (t1-t0)", t[:, 1] - t[:, 0])
    stat("first k-tile landed (t2-t1)", t[:, 2] - t[:, 1])
    stat(f"main loop (t3-t2), {nk} k-steps", t[:, 3] - t[:, 2])
    stat("  per k-step", (t[:, 3] - t[:, 2]) / max(nk, 1))
    stat("epilogue issue (t4-t3)", t[:, 4] - t[:, 3])
    stat("store drain (t5-t4)", t[:, 5] - t[:, 4])
    stat("workgroup life (t5-t0)", life)
    return rt0.min(), rt1.max()


buf2 = torch.zeros_like(buf)
# two dependent launches back to back (the second one reads nothing from the first, but stream order serialises them)
torch.cuda.synchronize()
assert lib.vlt5dbg_set_timeline(C.c_void_p(buf.data_ptr())) == 0
run()
assert lib.vlt5dbg_set_timeline(C.c_void_p(buf2.data_ptr())) == 0     # hipMemcpyToSymbol: synchronises with the first launch
run()
torch.cuda.synchronize()
# same pair captured in a graph is not possible (the symbol copy), so measure the launch-to-launch gap with one buffer:
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    with torch.cuda.graph(g, stream=st):
        for _ in range(4):
            run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.replay(); torch.cuda.synchronize()
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f"M={M} N={N} K={K} akm={akm} bkm={bkm} tile={tm}x{tn} f32={int(f32)}: {ntiles} workgroups; graph replay of 4 launches: "
      f"{e0.elapsed_time(e1) * 250:.1f} us per launch")
t = buf2.cpu().numpy().astype(np.uint64).reshape(ntiles, 8)
s0, s1 = analyse(t, "single launch")
print(f"  kernel span (first start -> last end, device clock): {(s1 - s0) / 100.0:.2f} us")
