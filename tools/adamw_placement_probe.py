#!/usr/bin/env python3
"""Is the AdamW pass's time a property of the ALLOCATION its streams got?  Eight independent sets of (p, g, m, v, bf16 shadow), each timed
three times; then mixed sets (streams taken from the fastest and the slowest set) to see whether single buffers carry the difference.

    python tools/adamw_placement_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqacl_amd._lib import lib, ptr, stream_ptr  # noqa: E402

dev = torch.device("cuda")
n = 225_722_368
tot = torch.ones(1, device=dev)


def timed(bufs, reps=6):
    p, g, m, v, pb = bufs
    t = [3]

    def step():
        t[0] += 1
        lib().vlt5_adamw_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(pb), n, 1e-4, 0.9, 0.999, 1e-6, 0.01, t[0], ptr(tot), 5.0, 1, stream_ptr())
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        step()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


sets = []
for s in range(8):
    bufs = [torch.empty(n, device=dev) for _ in range(4)] + [torch.empty(n, device=dev, dtype=torch.bfloat16)]
    bufs[0].normal_(); bufs[1].normal_().mul_(1e-3); bufs[2].zero_(); bufs[3].zero_()
    sets.append(bufs)
res = []
for s, bufs in enumerate(sets):
    ts = [timed(bufs) for _ in range(3)]
    res.append(sum(ts) / 3)
    print(f"set {s}: {ts[0]:7.1f} {ts[1]:7.1f} {ts[2]:7.1f} us   ({30 * n / res[-1] / 1e6:5.2f} TB/s)   p at {sets[s][0].data_ptr():#x}", flush=True)
fast, slow = min(range(8), key=lambda i: res[i]), max(range(8), key=lambda i: res[i])
print(f"fastest set {fast} ({res[fast]:.1f} us), slowest set {slow} ({res[slow]:.1f} us)")
names = ["p", "g", "m", "v", "pb"]
for k in range(5):
    mixed = list(sets[fast])
    mixed[k] = sets[slow][k]
    print(f"fastest set with the slowest set's {names[k]:2s}: {timed(mixed):7.1f} us")
for k in range(5):
    mixed = list(sets[slow])
    mixed[k] = sets[fast][k]
    print(f"slowest set with the fastest set's {names[k]:2s}: {timed(mixed):7.1f} us")
# one stream at a time: a read-only / write-only pass over each buffer of the two sets (torch sum / fill), GB/s
for tag, s in (("fastest", fast), ("slowest", slow)):
    line = []
    for k in range(4):
        b = sets[s][k]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b.sum(); e0.record()
        for _ in range(5):
            b.sum()
        e1.record(); e1.synchronize()
        line.append(f"{names[k]} read {4 * n * 5 / e0.elapsed_time(e1) / 1e6:5.2f}")
    print(f"{tag} set, single-stream reads (TB/s): " + "  ".join(line))
