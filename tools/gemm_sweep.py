#!/usr/bin/env python3
"""Time every GEMM shape of one train step under each tile / split-K choice (HIP events) -> gpurun_out/gemm_sweep.json.
Used to derive the tile heuristic in csrc/gemm.hip.  Run on the GPU box: python tools/gemm_sweep.py [--batch 80]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import gemm_schedule  # noqa: E402
from vqacl_amd import VLT5Config, ops  # noqa: E402


def timed(fn, reps=8):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def timed_graph(fn, reps=20):
    """Same, but the launches are replayed from a HIP graph: no host launch cost between kernels (small GEMMs take less
    GPU time than one ctypes call)."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


def main():
    B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 80
    auto_only = "--auto-only" in sys.argv            # only the heuristic's choice (for A/B runs of kernel variants)
    torch_ref = "--torch-ref" in sys.argv            # also time torch.matmul (hipBLASLt/rocBLAS) on the same operands
    tm = timed_graph if "--graph" in sys.argv else timed
    dev = torch.device("cuda")
    cfg = VLT5Config()
    BF = torch.bfloat16
    res = []
    seen = set()
    for count, batch, M, N, K, akm, bkm, of32 in gemm_schedule(cfg, B, 20, 36, 5):
        count = count * batch
        key = (M, N, K, akm, bkm)
        if key in seen:
            continue
        seen.add(key)
        A = torch.randn((K, M) if akm else (M, K), device=dev).to(BF)
        Bm = torch.randn((K, N) if bkm else (N, K), device=dev).to(BF)
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if of32 else BF)
        row = dict(M=M, N=N, K=K, akm=akm, bkm=bkm, count=count, gflop=2.0 * M * N * K / 1e9, t={})
        # (the heuristic's own choice is timed LAST: the first timing of a shape also pays the first touch of a freshly allocated output --
        # +9 us on the 171 MB output of the stacked cross-K/V projection)
        for tile in (((0, 0),) if auto_only else ((256, 256), (224, 256), (160, 256), (128, 128), (128, 64), (64, 128), (64, 64), (0, 0))):
            if akm and tile[0] in (224, 160):
                continue
            splits = (1, 2, 4, 8) if (akm and bkm and of32 and M * N % 1 == 0 and K >= 1024) else (1,)
            for sk in splits:
                if sk > 1 and tile == (0, 0):
                    continue
                us = tm(lambda: ops.gemm(A, Bm, M, N, K, a_kmajor=bool(akm), b_kmajor=bool(bkm), out=out, tile=tile, split_k=sk))
                row["t"][f"{tile[0]}x{tile[1]}/sk{sk}"] = round(us, 1)
        best = min((v, k) for k, v in row["t"].items() if auto_only or not k.startswith("0x0"))
        if torch_ref:
            At = A.t() if akm else A                                      # logical [M,K]
            Bt = Bm if bkm else Bm.t()                                    # logical [K,N]
            o2 = torch.empty(M, N, device=dev, dtype=BF)
            row["t"]["torch"] = round(tm(lambda: torch.matmul(At, Bt, out=o2)), 1)
        row["best"] = best[1]
        row["best_us"] = best[0]
        row["best_tflops"] = round(row["gflop"] / best[0] * 1e3, 1)
        res.append(row)
        print(f"M={M:6d} N={N:6d} K={K:6d} akm={akm} bkm={bkm} x{count:3d}  auto {row['t']['0x0/sk1']:7.1f}us  best {best[1]:>12s} {best[0]:7.1f}us "
              f"{row['best_tflops']:6.1f} TF | " + " ".join(f"{k}={v}" for k, v in row["t"].items()), flush=True)
    tot_auto = sum(r["count"] * r["t"]["0x0/sk1"] for r in res) / 1e3
    tot_best = sum(r["count"] * r["best_us"] for r in res) / 1e3
    print(f"GEMM ms/step: auto {tot_auto:.2f}  best-per-shape {tot_best:.2f}")
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/gemm_sweep.json", "w"), indent=1)


if __name__ == "__main__":
    main()
