#!/usr/bin/env python3
"""What does an event cost inside a chain of dependent kernels?  A chain of 200 small GEMM launches (a) bare, (b) with an event
recorded after every 8th launch, (c) recorded + waited for by a second stream (nothing else enqueued there), (d) recorded + waited
for + a tiny kernel on the second stream.  Guides how many per-bucket events the backward may afford."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqacl_amd import ops

dev = torch.device("cuda")
A = torch.randn(4480, 768, device=dev).to(torch.bfloat16)
W = torch.randn(768, 768, device=dev).to(torch.bfloat16)
out = torch.empty(4480, 768, device=dev, dtype=torch.bfloat16)
side = torch.cuda.Stream()
tiny = torch.zeros(64, device=dev)
N, every = 200, 8


def chain(mode, blocking_events=False):
    evs = [torch.cuda.Event(enable_timing=False, blocking=blocking_events) for _ in range(N // every + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        ops.gemm(A, W, 4480, 768, 768, out=out)
        if mode and i % every == every - 1:
            e = evs[i // every]
            e.record()
            if mode >= 2:
                side.wait_event(e)
            if mode >= 3:
                with torch.cuda.stream(side):
                    tiny.add_(1.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


for mode, name in ((0, "bare chain"), (1, "+ record"), (2, "+ record + wait on 2nd stream"), (3, "+ record + wait + tiny kernel on 2nd stream")):
    for _ in range(2):
        chain(mode)
    ts = sorted(chain(mode) for _ in range(5))
    print(f"{name:48s} {ts[2]:9.1f} us per {N} launches  ({(ts[2]) / N:6.2f} us/launch)", flush=True)
base = sorted(chain(0) for _ in range(5))[2]
for mode in (1, 2, 3):
    t = sorted(chain(mode) for _ in range(5))[2]
    print(f"mode {mode}: extra per event {(t - base) / (N // every):7.2f} us")
