import os, sys, torch
sys.path.insert(0, os.getcwd())
from vqacl_amd import ops
BF = torch.bfloat16
dev = torch.device("cuda")
M, N, K = 400, 768, 768
g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
A = (torch.randn(M, K, generator=g)).to(BF)
B = (torch.randn(N, K, generator=g) + torch.arange(N)[:, None] * 0.01).to(BF)
ref = A.float() @ B.float().t()
for tile in [(256, 256), (224, 256), (128, 128), (64, 64)]:
    for rep in range(3):
        outb = ops.gemm(A.to(dev), B.to(dev), M, N, K, tile=tile).float().cpu()
        bad = ((outb - ref).abs() > 0.05 + 0.01 * ref.abs()).nonzero()
        print(tile, rep, len(bad), bad[:3].tolist(), bad[-3:].tolist() if len(bad) else "", [float(outb[r, c]) for r, c in bad[:3].tolist()])
