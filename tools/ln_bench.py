#!/usr/bin/env python3
"""HIP-event timing of the LayerNorm kernels at the model's shapes against their HBM roofline (algorithmic bytes / time).
usage: python tools/ln_bench.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqacl_amd._lib import check, lib, ptr, stream_ptr   # noqa: E402

BF = torch.bfloat16


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    L = lib()
    for rows, d, nsl in ((4480, 768, 1), (400, 768, 1), (400, 768, 8), (1792, 1024, 1)):
        x = torch.randn(rows, d, device=dev)
        w = torch.ones(d, device=dev)
        dy = torch.randn(max(nsl, 1), rows, d, device=dev)
        dx = torch.zeros(rows, d, device=dev)
        yb = torch.empty(rows, d, device=dev, dtype=BF)
        dxb = torch.empty(rows, d, device=dev, dtype=BF)
        rstd = torch.empty(rows, device=dev)
        xo = torch.empty(rows, d, device=dev)
        part = torch.empty(L.vlt5_layernorm_bwd_blocks(rows), d, device=dev)
        st = stream_ptr()
        t_f = timeit(lambda: check(L.vlt5_layernorm_fwd(ptr(x), ptr(w), ptr(yb), None, ptr(rstd), rows, d, 1e-6, 0.0, 0, 0, 0, st)))
        b_f = rows * d * 6
        t_fs = timeit(lambda: check(L.vlt5_layernorm_fwd_slabs(ptr(dy), nsl, rows * d, ptr(x), ptr(xo), 0.1, 7, ptr(w), ptr(yb), None,
                                                              ptr(rstd), rows, d, 1e-6, 0.0, 0, 0, 0, st)))
        b_fs = rows * d * (4 * nsl + 4 + 4 + 2)
        t_b = timeit(lambda: check(L.vlt5_layernorm_bwd_slabs(ptr(dy), nsl, rows * d, ptr(x), ptr(w), ptr(rstd), ptr(dx), None, ptr(part),
                                                              rows, d, 1, 0, 0.0, 0, 0, 0, ptr(dxb), 0.1, 9, st)))
        b_b = rows * d * (4 * nsl + 4 + 4 + 4 + 2)
        print(f"rows={rows:5d} d={d:4d} slabs={nsl}:  fwd {t_f:6.2f} us {b_f / t_f / 1e3:7.1f} GB/s | fwd+slabs+resid {t_fs:6.2f} us "
              f"{b_fs / t_fs / 1e3:7.1f} GB/s | bwd(accum, bf16 out) {t_b:6.2f} us {b_b / t_b / 1e3:7.1f} GB/s")


if __name__ == "__main__":
    main()
