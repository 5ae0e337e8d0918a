#!/usr/bin/env python3
"""Why are the GEMMs 20-40 % slower inside a step than in a back-to-back replay?  Each shape is timed (event pair attached to the
dispatch) warm, after a cache flush (1 GB fill: L2 + Infinity Cache evicted), and after a flush followed by re-touching only the
weight operand, only the activation operand, or both -- the decomposition says what a prefetch into the Infinity Cache could buy."""
import ctypes as C
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqacl_amd import ops
from vqacl_amd._lib import GemmTimingRec, lib, stream_ptr

dev = torch.device("cuda")
BF = torch.bfloat16
junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
sink = torch.zeros(1, device=dev)


def touch(t):
    sink.add_(t.view(torch.int16).sum().float())     # streams the tensor through the Infinity Cache / L2


def run(M, N, K, akm, bkm, of32, gate=False, resid=False, reps=6):
    A = torch.randn((K, M) if akm else (M, K), device=dev).to(BF)
    Bm = torch.randn((K, N) if bkm else (N, K), device=dev).to(BF)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if of32 else BF)
    g_t = torch.randn(M, N, device=dev).to(BF) if gate else None
    r_t = torch.randn(M, N, device=dev) if resid else None
    g, _, keep = ops.gemm_desc(A, Bm, M, N, K, a_kmajor=bool(akm), b_kmajor=bool(bkm), out=out, gate=g_t, resid=r_t)
    fn, gp = lib().vlt5_gemm_bf16, C.byref(g)
    res = {}
    for mode in ("warm", "cold", "cold+B", "cold+A", "cold+A+B", "cold+all"):
        assert lib().vlt5_gemm_timing_enable(64) == 0
        fn(gp, stream_ptr())
        for _ in range(reps):
            if mode != "warm":
                junk.fill_(1)
                if "B" in mode or "all" in mode:
                    touch(Bm)
                if "+A" in mode or "all" in mode:
                    touch(A)
                if "all" in mode:
                    touch(out.view(torch.int16) if not of32 else out.view(torch.int32).view(torch.int16))
                    if g_t is not None:
                        touch(g_t)
                    if r_t is not None:
                        touch(r_t.view(torch.int16))
            fn(gp, stream_ptr())
        torch.cuda.synchronize()
        recs = (GemmTimingRec * 64)()
        n = lib().vlt5_gemm_timing_collect(recs, 64)
        lib().vlt5_gemm_timing_enable(0)
        ts = sorted(r.ms for r in recs[1:n])
        res[mode] = ts[len(ts) // 2] * 1e3
    tag = f"M={M} N={N} K={K} akm={akm} bkm={bkm} f32={of32} gate={int(gate)} resid={int(resid)}"
    print(f"{tag:64s} " + "  ".join(f"{k} {v:6.1f}" for k, v in res.items()), flush=True)


run(4480, 3072, 768, 0, 0, 0)                 # FFN wi forward
run(4480, 3072, 768, 0, 1, 0, gate=True)      # FFN hidden gradient (gate by the saved activation)
run(4480, 768, 3072, 0, 1, 1)                 # FFN wi dgrad
run(4480, 768, 3072, 0, 0, 1, resid=True)     # FFN wo forward (+ residual)
run(4480, 768, 2304, 0, 1, 1)                 # q|k|v dgrad
run(4480, 768, 768, 0, 0, 1, resid=True)      # attention output projection
run(4480, 768, 768, 0, 1, 0)                  # its dgrad
run(3072, 768, 4480, 1, 1, 1)                 # a weight gradient (one layer)
run(400, 768, 768, 0, 0, 1, resid=True)       # decoder-sized
run(400, 3072, 768, 0, 0, 0)
