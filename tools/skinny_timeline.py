#!/usr/bin/env python3
"""Per-workgroup phase timeline of the row-panel GEMM (debug hook vlt5dbg_skinny_timeline): usage  make -C vqacl_amd/csrc exp; VLT5_LIB=vqacl_amd/libvlt5_exp.so python tools/skinny_timeline.py "M N K ln rows cols" ..."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.experiments import skinny_py as ops_sk
from vqacl_amd import ops
from vqacl_amd._lib import lib, ptr
dev = torch.device("cuda")
L = lib()
L.vlt5dbg_skinny_timeline.argtypes = [C.c_void_p]
for spec in sys.argv[1:]:
    M, N, K, ln, rows, cols = [int(v) for v in spec.split()]
    X = torch.randn(M, K, device=dev)
    lw = torch.rand(K, device=dev) + 0.5
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    kw = dict(A=None if ln else X.to(torch.bfloat16), ln_x=X if ln else None, ln_w=lw if ln else None, panel_rows=rows, chunk_cols=cols)
    for _ in range(3):
        ops_sk.skinny_gemm(W, M, N, K, **kw)
    buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    L.vlt5dbg_skinny_timeline(ptr(buf))
    torch.cuda.synchronize()
    ops_sk.skinny_gemm(W, M, N, K, **kw)
    torch.cuda.synchronize()
    L.vlt5dbg_skinny_timeline(None)
    t = buf.view(-1, 8).cpu()
    t = t[t[:, 7] != 0].double()
    n = t.shape[0]
    names = ["setup+requests", "prologue (panel)", "barrier", "main loop", "epilogue+stores land"]
    print(f"M={M} N={N} K={K} ln={ln} panel {rows}x{cols}: {n} workgroups; span of the launch {float((t[:, 6].max() - t[:, 7].min()) / 100):.2f} us "
          f"(first start -> last end, 100 MHz clock); workgroup life mean {float((t[:, 6] - t[:, 7]).mean() / 100):.2f} us, max {float((t[:, 6] - t[:, 7]).max() / 100):.2f} us; "
          f"start skew {float((t[:, 7].max() - t[:, 7].min()) / 100):.2f} us")
    for i, nm in enumerate(names):
        d = t[:, i + 1] - t[:, i]
        print(f"    {nm:24s} mean {float(d.mean()):8.0f} clk   min {float(d.min()):8.0f}   max {float(d.max()):8.0f}")
