#!/usr/bin/env python3
"""Which kernels does hipBLASLt (through torch.matmul) pick for the model's GEMM shapes?  Run under
`rocprofv3 --kernel-trace` and read the kernel names (macro tile, wave layout, depth) from the trace:
    rocprofv3 --kernel-trace -d out -o r -- python3 tools/blaslt_probe.py ; python3 tools/rocpd_stats.py out/.../r_results.db"""
import torch

dev = torch.device("cuda:0")
BF = torch.bfloat16
shapes = [(4480, 2304, 768, "nt"), (4480, 768, 768, "nt"), (4480, 3072, 768, "nt"), (4480, 768, 3072, "nt"),
          (4480, 768, 2304, "nn"), (4480, 768, 3072, "nn"), (4480, 3072, 768, "nn"),
          (2304, 768, 4480, "tn"), (3072, 768, 4480, "tn"), (400, 768, 768, "nt"), (400, 768, 3072, "nt")]
for M, N, K, mode in shapes:
    if mode == "nt":      # y = x @ w.T
        a = torch.randn(M, K, device=dev).to(BF); b = torch.randn(N, K, device=dev).to(BF)
        f = lambda: a @ b.t()
    elif mode == "nn":    # dx = dy @ w
        a = torch.randn(M, K, device=dev).to(BF); b = torch.randn(K, N, device=dev).to(BF)
        f = lambda: a @ b
    else:                 # dw = dy.T @ x
        a = torch.randn(K, M, device=dev).to(BF); b = torch.randn(K, N, device=dev).to(BF)
        f = lambda: a.t() @ b
    for _ in range(6):
        f()
    torch.cuda.synchronize()
    # a marker kernel between shapes so the trace can be split
    torch.zeros(M + 1, device=dev).add_(1)
torch.cuda.synchronize()
