#!/usr/bin/env python3
"""Which hipBLASLt kernels torch.matmul picks for the GEMM shapes of the step where the vendor library is ahead (run under
rocprofv3 --kernel-trace --stats; the kernel names carry the macro tile MT, the split GSU and the workgroup shape)."""
import torch
dev = torch.device("cuda")
shapes = [(4480, 768, 3072), (4480, 3072, 768), (4480, 768, 768), (4480, 2304, 768), (4640, 18432, 768), (400, 768, 3072), (400, 768, 32200), (400, 3072, 768)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        c = a @ b.t()
    e1.record()
    torch.cuda.synchronize()
    print(f"M={M} N={N} K={K}: {e0.elapsed_time(e1) / 50 * 1000:.1f} us", flush=True)
