#!/usr/bin/env python3
"""EXPERIMENT, not product (needs tools/experiments/ksplit_kernel.patch applied and the library rebuilt): measured in round 6 and not kept
-- profiles/r06_d_ksplit_probe.txt, r06_d_ab_ksplit.txt.
The K-split kernel (csrc/ksplit.hip, vlt5_tuning.gemm_ksplit = 2) against the tiled kernel on the decoder's forward projections:
correctness against an f32 torch product of the same bf16 operands (and bit-equality of the dropout mask with the tiled kernel's),
graph-replayed timing warm (one weight matrix) and cold-ish (rotating through 48 weight matrices, > Infinity Cache).

    python tools/ksplit_probe.py ["M N K relu f32out" ...]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_sweep import timed_graph  # noqa: E402
from vqacl_amd import _lib as L  # noqa: E402
from vqacl_amd import ops  # noqa: E402
from vqacl_amd._lib import lib, stream_ptr  # noqa: E402

dev = torch.device("cuda")
BF = torch.bfloat16
specs = [a for a in sys.argv[1:] if not a.startswith("--")] or [
    "400 2304 768 0 0", "400 768 768 0 0", "400 3072 768 1 0", "400 768 768 0 1", "20 768 768 0 1", "80 3072 768 1 0", "500 2304 768 0 0",
    "64 4096 1024 1 0", "160 1024 1024 0 1"]
ROT = 48
torch.manual_seed(0)
on, off = L.make_tuning(gemm_ksplit=True), L.make_tuning(gemm_ksplit=False)
fn = lib().vlt5_gemm_bf16
for spec in specs:
    M, N, K, relu, f32 = [int(v) for v in spec.split()]
    A = (torch.randn(M, K, device=dev) * 1.5).to(BF)
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(BF) for _ in range(ROT)]
    R = torch.randn(M, N, device=dev) if f32 else None
    ref = A.float() @ Ws[0].float().t()
    if relu:
        ref = ref.relu()
    outs = {}
    for name, tun in (("ksplit", on), ("tiled", off)):
        for dp in (0.0, 0.1):
            g, out, keep = ops.gemm_desc(A, Ws[0], M, N, K, relu=bool(relu), resid=R, out_f32=bool(f32), drop_p=dp, drop_seed=1234)
            g.tuning = C.pointer(tun)
            rc = fn(C.byref(g), stream_ptr())
            assert rc == 0, rc
            torch.cuda.synchronize()
            outs[name, dp] = out.float().clone()
        want = ref + R if f32 else ref
        err = float((outs[name, 0.0] - want).abs().max() / want.abs().max())
        descs = []
        o2 = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else BF)
        for W in Ws:
            g, _, keep = ops.gemm_desc(A, W, M, N, K, out=o2, relu=bool(relu), resid=R, drop_p=0.1, drop_seed=77)
            g.tuning = C.pointer(tun)
            descs.append((g, keep))
        warm = timed_graph(lambda: fn(C.byref(descs[0][0]), stream_ptr()))
        it = [0]

        def rot():
            fn(C.byref(descs[it[0] % ROT][0]), stream_ptr())
            it[0] += 1
        cold = timed_graph(rot, reps=ROT)
        print(f"M={M:4d} N={N:5d} K={K:5d} relu={relu} f32+resid={f32}  {name:6s}: err {err:.1e}  warm {warm:6.2f} us  rot {cold:6.2f} us", flush=True)
    # same dropout mask as the tiled kernel (the backward regenerates it from the same counters), values equal up to the summation order
    a, b = outs["ksplit", 0.1], outs["tiled", 0.1]
    base = R if f32 else torch.zeros_like(a)
    same_mask = bool((((a - base) == 0) == ((b - base) == 0)).float().mean() > 0.9999)
    close = float((a - b).abs().max() / b.abs().max())
    print(f"      dropout 0.1: zero pattern equal to the tiled kernel's: {same_mask};  max |ksplit - tiled| / max |tiled| = {close:.1e}")
