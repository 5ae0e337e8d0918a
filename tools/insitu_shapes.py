#!/usr/bin/env python3
"""Per-SHAPE in-situ timing of every GEMM-family dispatch of real train steps (vlt5_gemm_timing_*: HIP events attached to the
dispatch): which launches of the step the GEMM time goes to, with the tile the dispatcher picked.  `bench.py` reports the same
records summed per instantiation; this splits them by (M, N, K, batch, operand orders)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch
from vqacl_amd import VLT5VQA, VLT5Config, FusedAdamW, reference_param_groups
from vqacl_amd._lib import GemmTimingRec, lib

large = "--large" in sys.argv                  # VL-T5-large (BASELINE configs[4]): d = 1024, 16 heads, d_ff = 4096, 24 layers, B = 32
args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else (32 if large else 80)
steps = 6
dev = torch.device("cuda")
kw = dict(d_model=1024, num_heads=16, d_ff=4096, num_layers=24) if large else {}
model = VLT5VQA(VLT5Config(dropout_rate=0.1, **kw), device=dev)
model.train()
opt = FusedAdamW(reference_param_groups(model, 0.01), model, lr=1e-4, eps=1e-6, max_grad_norm=5.0)
batch = {k: v.to(dev) for k, v in synthetic_batch(B, seed=1).items()}


def step():
    res = model.train_step(batch, 0, 0.5, 0.3)
    res["loss"].backward()
    opt.step()
    for p in model.parameters():
        p.grad = None


for _ in range(4):
    step()
cap = 1024 * steps
assert lib().vlt5_gemm_timing_enable(cap) == 0
torch.cuda.synchronize()
for _ in range(steps):
    step()
torch.cuda.synchronize()
recs = (GemmTimingRec * cap)()
n = lib().vlt5_gemm_timing_collect(recs, cap)
lib().vlt5_gemm_timing_enable(0)
by = {}
for r in recs[:n]:
    key = (r.tile_m, r.tile_n, r.M, r.N, r.K, r.batch, r.a_kmajor, r.b_kmajor, r.splits, r.workgroups, r.out_f32, r.M2, r.N2, r.K2, r.batch2)
    v = by.setdefault(key, [0, 0.0])
    v[0] += 1
    v[1] += r.ms
tot = sum(v[1] for v in by.values()) / steps
print(f"# {n / steps:.0f} dispatches per step, {tot:.3f} ms per step in the GEMM family")
print("tile      M      N      K  batch akm bkm splits   wgs f32  calls/step   avg_us   ms/step   TFLOP/s")
for key, v in sorted(by.items(), key=lambda kv: -kv[1][1]):
    tm, tn, M, N, K, bt, akm, bkm, sp, wg, f32, M2, N2, K2, bt2 = key
    us = v[1] / v[0] * 1e3
    tf = 2.0 * (bt * M * N * K + bt2 * M2 * N2 * K2) / (us * 1e-6) / 1e12            # (M2 x N2: the second problem of a grouped launch)
    print(f"{tm:3d}x{tn:<3d} {M:6d} {N:6d} {K:6d} {bt:5d} {akm:3d} {bkm:3d} {sp:6d} {wg:5d} {f32:3d} {v[0] / steps:10.1f} {us:9.2f} {v[1] / steps:9.3f} {tf:9.1f}" + (f"   + {M2}x{N2}x{K2} x{bt2} grouped" if M2 else ""))
