#!/usr/bin/env python3
"""Does the AdamW pass depend on WHERE its streams lie?  (round 6: its time varies 1.13-1.46 ms per step between boxes with the same code,
most of the box-to-box spread of the step.)  The kernel reads p, g, m, v and writes p, m, v, bf16(p) -- eight streams advancing in lockstep.
Carves them out of ONE allocation with a chosen byte offset between consecutive streams (on top of the 226 M-element stride) and times
the pass for each; also a two-stream copy of the same byte count as the box's own yardstick.

    python tools/adamw_layout_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqacl_amd._lib import lib, ptr, stream_ptr  # noqa: E402

dev = torch.device("cuda")
n = 225_722_368                      # the base model's flat buffer (multiple of 64)
MAXPAD = 8 << 20
raw = torch.empty(4 * (4 * n + MAXPAD) + 2 * n + MAXPAD + 4096, device=dev, dtype=torch.uint8)
tot = torch.ones(1, device=dev)


def carve(pad):
    off, out = 0, []
    for i in range(4):
        out.append(raw[off:off + 4 * n].view(torch.float32))
        off += 4 * n + pad
    out.append(raw[off:off + 2 * n].view(torch.bfloat16))
    return out


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


a = torch.empty(15 * n // 4, device=dev)          # 15 B/param read + 15 written = the AdamW byte count as a plain copy
b = torch.empty_like(a)
ms = timed(lambda: b.copy_(a))
print(f"copy of the same byte count (2 streams): {ms * 1e3:7.1f} us  {30 * n / ms / 1e9:5.2f} TB/s")
for pad in (0, 256, 4096, 4096 + 256, 65536, 65536 + 4096, 1 << 20, (1 << 20) + 65536 + 4096, 2 << 20, (2 << 20) + 4096, 3 * (1 << 19) + 8192):
    p, g, m, v, pb = carve(pad)
    p.normal_(); g.normal_().mul_(1e-3); m.zero_(); v.zero_()
    t = [3]

    def step():
        t[0] += 1
        lib().vlt5_adamw_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(pb), n, 1e-4, 0.9, 0.999, 1e-6, 0.01, t[0], ptr(tot), 5.0, 1, stream_ptr())
    ms = timed(step)
    print(f"offset between streams = 4n + {pad:8d} B: adamw {ms * 1e3:7.1f} us  {30 * n / ms / 1e9:5.2f} TB/s", flush=True)
# the model's own layout: separate allocations, as FusedAdamW / VLT5 make them
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 1e-3; m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
pb = torch.empty(n, device=dev, dtype=torch.bfloat16)
t = [3]


def step2():
    t[0] += 1
    lib().vlt5_adamw_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(pb), n, 1e-4, 0.9, 0.999, 1e-6, 0.01, t[0], ptr(tot), 5.0, 1, stream_ptr())
ms = timed(step2)
print(f"separate torch allocations (addresses mod 2 MiB: {[hex(x.data_ptr() % (2 << 20)) for x in (p, g, m, v, pb)]}): adamw {ms * 1e3:7.1f} us  {30 * n / ms / 1e9:5.2f} TB/s")

# separate allocations again, each stream starting `k * stagger` bytes into its own (over-sized) allocation: is it the common 2 MiB phase?
for stagger in (0, 256, 4096, 16384, 65536, 262144, 1 << 20, 1105920):
    bufs = [torch.empty(4 * n + 8 * (2 << 20), device=dev, dtype=torch.uint8) for _ in range(5)]
    views = []
    for k, bfr in enumerate(bufs):
        o = k * stagger
        views.append(bfr[o:o + (4 * n if k < 4 else 2 * n)].view(torch.float32 if k < 4 else torch.bfloat16))
    p, g, m, v, pb = views
    p.normal_(); g.normal_().mul_(1e-3); m.zero_(); v.zero_()
    t = [3]

    def step3():
        t[0] += 1
        lib().vlt5_adamw_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(pb), n, 1e-4, 0.9, 0.999, 1e-6, 0.01, t[0], ptr(tot), 5.0, 1, stream_ptr())
    ms = timed(step3)
    print(f"separate allocations, stream k starts k x {stagger:8d} B into its own: adamw {ms * 1e3:7.1f} us  {30 * n / ms / 1e9:5.2f} TB/s   "
          f"(bases mod 2 MiB: {[hex(x.data_ptr() % (2 << 20)) for x in views]})", flush=True)
    del bufs, views, p, g, m, v, pb
    torch.cuda.empty_cache()
# the copy yardstick with the destination staggered
a = torch.empty(15 * n // 4 + (1 << 20), device=dev)
b = torch.empty(15 * n // 4 + (1 << 20), device=dev)
for so in (0, 1024, 16384, 276480):
    bb = b[so:so + 15 * n // 4]
    aa = a[:15 * n // 4]
    ms = timed(lambda: bb.copy_(aa))
    print(f"copy, destination starts {4 * so:8d} B into its allocation: {ms * 1e3:7.1f} us  {30 * n / ms / 1e9:5.2f} TB/s")
