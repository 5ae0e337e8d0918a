#!/usr/bin/env python3
"""The AdamW pass over streams carved out of ONE allocation, against the byte offset between consecutive streams (large pads: is there a
period in the physical address hashing that decides how the lock-stepped streams collide?).  python tools/adamw_pad_probe.py [repeat]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqacl_amd._lib import lib, ptr, stream_ptr  # noqa: E402

dev = torch.device("cuda")
n = 225_722_368
MB = 1 << 20
tot = torch.ones(1, device=dev)
pads = [0, 1 * MB, 2 * MB, 3 * MB, 4 * MB, 6 * MB, 8 * MB, 12 * MB, 16 * MB, 24 * MB, 32 * MB, 48 * MB, 64 * MB, 96 * MB, 128 * MB, 192 * MB, 256 * MB,
        5 * MB + 4096, 37 * MB, 101 * MB]
raw = torch.empty(4 * (4 * n + max(pads)) + 2 * n + 4096, device=dev, dtype=torch.uint8)
print(f"arena at {raw.data_ptr():#x}; stream stride 4n = {4 * n} B = {4 * n / MB:.3f} MiB (4n mod 2 MiB = {4 * n % (2 * MB)})")


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


for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    for pad in pads:
        off, bufs = 0, []
        for i in range(4):
            bufs.append(raw[off:off + 4 * n].view(torch.float32))
            off += 4 * n + pad
        bufs.append(raw[off:off + 2 * n].view(torch.bfloat16))
        bufs[0].normal_(); bufs[1].normal_().mul_(1e-3); bufs[2].zero_(); bufs[3].zero_()
        us = timed(bufs)
        print(f"pad {pad / MB:8.3f} MiB (stride mod 64 MiB = {(4 * n + pad) % (64 * MB) / MB:7.3f}): {us:7.1f} us  {30 * n / us / 1e6:5.2f} TB/s", flush=True)
