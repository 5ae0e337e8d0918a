import sys, torch
sys.path.insert(0, '.')
from vqacl_amd import ops
dev = torch.device('cuda')
BF = torch.bfloat16
def run(tag, **kw):
    print(tag, flush=True)
    out = ops.gemm(**kw)
    torch.cuda.synchronize()
    print('   ok', float(out.float().abs().mean()), flush=True)
M, N, K = 448, 512, 256
A = torch.randn(M, K, device=dev).to(BF); B = torch.randn(N, K, device=dev).to(BF)
Bk = B.t().contiguous(); Ak = A.t().contiguous()
for tile in [(64, 64), (128, 64), (64, 128), (128, 128), (160, 256), (224, 256), (256, 256)]:
    run(f"plain rm/rm {tile}", A=A, B=B, M=M, N=N, K=K, tile=tile)
    run(f"plain rm/km {tile}", A=A, B=Bk, M=M, N=N, K=K, b_kmajor=True, tile=tile)
    if tile[0] not in (160, 224):
        run(f"plain km/km {tile}", A=Ak, B=Bk, M=M, N=N, K=K, a_kmajor=True, b_kmajor=True, out_f32=True, tile=tile)
    run(f"relu+drop {tile}", A=A, B=B, M=M, N=N, K=K, relu=True, drop_p=0.1, drop_seed=3, tile=tile)
    h = ops.gemm(A, B, M, N, K, relu=True, tile=tile)
    run(f"gate rm/km {tile}", A=A, B=Bk, M=M, N=N, K=K, b_kmajor=True, gate=h, gate_scale=1.1, tile=tile)
    bits = torch.zeros(M, N // 8, device=dev, dtype=torch.uint8)
    run(f"relu bits {tile}", A=A, B=B, M=M, N=N, K=K, relu=True, drop_p=0.1, drop_seed=3, tile=tile, relu_bits_out=bits)
    run(f"gate bits {tile}", A=A, B=Bk, M=M, N=N, K=K, b_kmajor=True, gate_bits=bits, gate_scale=1.1, tile=tile)
print("ALL OK")
