#!/usr/bin/env python3
"""Step-faithful GEMM sweep: every GEMM-family launch of ONE real train step, exactly as the engine launches it -- batched over layers
(grid.z), grouped with its second problem, cut along K where the engine cuts it -- replayed warm from a HIP graph under every tile /
split choice and against torch (hipBLASLt / rocBLAS: matmul, or bmm for a layer batch) on the same logical problem.

Where tools/gemm_sweep.py times each distinct (M, N, K) once as a single un-split launch -- which is NOT how the step runs the weight
gradients (one launch per weight kind over 6 or 12 layers, some with a second problem in the same grid) or the decoder's sublayer
outputs (split-K slabs summed by the norm behind them) -- this tool takes the launch list from the engine's own dispatch records
(vlt5_gemm_timing_*), so "auto" IS the step's configuration and "best" / "torch" are alternatives for the same launch.

    python tools/gemm_step_sweep.py [--batch 80] [--large] [--quick]

Columns: in-step (event-timed inside real steps, cold operands), auto (the step's configuration, warm replay), best alternative,
torch.  Footer: GEMM ms per step auto / best-per-launch / torch-where-it-wins, and the launches more than 3 % behind torch.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch  # noqa: E402
from vqacl_amd import VLT5VQA, VLT5Config, FusedAdamW, reference_param_groups  # noqa: E402
from vqacl_amd import _lib as L  # noqa: E402
from vqacl_amd._lib import GemmDesc, GemmTimingRec, lib, ptr, stream_ptr  # noqa: E402

BF = torch.bfloat16
TILES = ((256, 256), (224, 256), (160, 256), (128, 128), (128, 64), (64, 128), (64, 64))


def step_records(B, steps=4, large=False):
    dev = torch.device("cuda")
    kw = dict(d_model=1024, num_heads=16, d_ff=4096, num_layers=24) if large else {}        # VL-T5-large (BASELINE configs[4])
    model = VLT5VQA(VLT5Config(dropout_rate=0.1, **kw), device=dev)
    model.train()
    opt = FusedAdamW(reference_param_groups(model, 0.01), model, lr=1e-4, eps=1e-6, max_grad_norm=5.0)
    batch = {k: v.to(dev) for k, v in synthetic_batch(B, seed=1).items()}

    def step():
        model.train_step(batch, 0, 0.5, 0.3)["loss"].backward()
        opt.step()
        for p in model.parameters():
            p.grad = None
    for _ in range(3):
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
        key = (r.tile_m, r.tile_n, r.M, r.N, r.K, max(r.batch, 1), r.a_kmajor, r.b_kmajor, max(r.splits, 1), r.out_f32, r.M2, r.N2, r.K2, r.batch2)
        v = by.setdefault(key, [0, 0.0])
        v[0] += 1
        v[1] += r.ms
    del model, opt, batch
    torch.cuda.empty_cache()
    return {k: (v[0] / steps, v[1] / v[0] * 1e3) for k, v in by.items()}


def replay_us(fn, reps=20):
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


class Problem:
    """Operands of one (possibly layer-batched) GEMM of the step: A [batch][M,K] (or [K,M] k-major), B [batch][N,K] (or [K,N])."""

    def __init__(self, M, N, K, batch, akm, bkm, of32, dev):
        self.M, self.N, self.K, self.batch, self.akm, self.bkm, self.of32 = M, N, K, batch, akm, bkm, of32
        self.A = (torch.randn((batch, K, M) if akm else (batch, M, K), device=dev) * 0.05).to(BF)
        self.B = (torch.randn((batch, K, N) if bkm else (batch, N, K), device=dev) * 0.05).to(BF)
        self.out = torch.empty(batch, M, N, device=dev, dtype=torch.float32 if of32 else BF)

    def desc(self, tile=(0, 0), split=1, ws=None, defer=False):
        g = GemmDesc()
        g.A, g.B, g.C = ptr(self.A), ptr(self.B), ptr(self.out)
        g.M, g.N, g.K = self.M, self.N, self.K
        g.lda, g.ldb, g.ldc = self.A.stride(1), self.B.stride(1), self.N
        g.a_kmajor, g.b_kmajor, g.alpha, g.out_f32 = self.akm, self.bkm, 1.0, int(self.of32)
        g.tile_m, g.tile_n = tile
        g.batch = self.batch
        g.batch_stride_a, g.batch_stride_b, g.batch_stride_c = self.A.stride(0), self.B.stride(0), self.out.stride(0)
        if split > 1:
            g.split_k, g.workspace, g.defer_reduce = split, ptr(ws), int(defer)
        return g

    def torch_fn(self):
        A = self.A.transpose(1, 2) if self.akm else self.A                    # logical [batch, M, K]
        Bt = self.B if self.bkm else self.B.transpose(1, 2)                   # logical [batch, K, N]
        o = torch.empty(self.batch, self.M, self.N, device=self.A.device, dtype=BF)
        if self.batch == 1:
            return lambda: torch.matmul(A[0], Bt[0], out=o[0])
        return lambda: torch.bmm(A, Bt, out=o)


def main():
    large = "--large" in sys.argv
    B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else (32 if large else 80)
    quick = "--quick" in sys.argv
    dev = torch.device("cuda")
    recs = step_records(B, large=large)
    fn = lib().vlt5_gemm_bf16
    rows = []
    print(f"# {'VL-T5-large' if large else 'VL-T5-base'}: {sum(c for c, _ in recs.values()):.0f} GEMM-family launches per step at B = {B}; one line per distinct launch configuration", flush=True)
    for key, (calls, insitu_us) in sorted(recs.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        tm, tn, M, N, K, bt, akm, bkm, sp, f32, M2, N2, K2, bt2 = key
        if (tm, tn) == (128, 384):
            continue                        # the fused q|k|v + attention kernel: not a plain GEMM launch (tools/attn_bench.py)
        p1 = Problem(M, N, K, bt, akm, bkm, f32, dev)
        p2 = Problem(M2, N2, K2, max(bt2, 1), akm, bkm, 1, dev) if M2 else None
        ws = torch.empty(max(sp, 8) * M * N * 4, device=dev, dtype=torch.uint8) if (f32 and bt == 1 and not p2) else None
        # a cut launch of the step leaves its slabs to the consumer (a norm forward / backward that sums them); the weight gradients'
        # own split reduces in place -- the in-step records do not say which, the shapes do: k-major A = weight gradient
        defer = bool(sp > 1 and not akm)

        def launch(tile, split):
            g = p1.desc(tile, split, ws, defer)
            keep = [g]
            if p2 is not None:
                g2 = p2.desc()
                g.grouped_with = C.addressof(g2)
                keep.append(g2)
            gp, sp_ = C.byref(g), stream_ptr()
            if fn(gp, sp_) != 0:
                return None
            return replay_us(lambda: fn(gp, stream_ptr()))
        t = {}
        auto = launch((tm, tn), sp)
        cand_tiles = [x for x in TILES if not (akm and x[0] in (224, 160))]
        cand_splits = [1] if (ws is None) else sorted({1, 2, 4, 8, sp})
        if quick:
            cand_tiles = [x for x in cand_tiles if x != (tm, tn)][:3]
        for tile in cand_tiles:
            for s_ in cand_splits:
                if (tile, s_) == ((tm, tn), sp) or s_ > max(1, (K // 64) // 4):
                    continue
                us = launch(tile, s_)
                if us is not None:
                    t[f"{tile[0]}x{tile[1]}/sk{s_}"] = round(us, 1)
        t1 = replay_us(p1.torch_fn())
        t2 = replay_us(p2.torch_fn()) if p2 is not None else 0.0
        torch_us = t1 + t2
        best_k, best = min(t.items(), key=lambda kv: kv[1]) if t else ("-", auto)
        gf = 2.0 * (bt * M * N * K + (bt2 * M2 * N2 * K2 if M2 else 0)) / 1e9
        rows.append(dict(key=key, calls=calls, insitu=insitu_us, auto=auto, best=min(best, auto), best_k=best_k if best < auto else "auto", torch=torch_us))
        what = f"{M}x{N}x{K}" + (f" x{bt}" if bt > 1 else "") + (f" + {M2}x{N2}x{K2}" + (f" x{bt2}" if bt2 > 1 else "") if M2 else "")
        print(f"{what:44s} {'km' if akm else 'rm'}/{'km' if bkm else 'rm'} {'f32' if f32 else 'b16'} {tm}x{tn}/sk{sp}{'d' if defer else ''} x{calls:4.1f}  "
              f"in-step {insitu_us:7.1f}  auto {auto:7.1f} us {gf / auto * 1e3:6.0f} TF  best {rows[-1]['best_k']:>12s} {rows[-1]['best']:7.1f}  "
              f"torch {torch_us:7.1f}{' (two launches)' if p2 is not None else ''}  | " + " ".join(f"{k}={v}" for k, v in sorted(t.items(), key=lambda kv: kv[1])[:6]), flush=True)
        del p1, p2, ws
        torch.cuda.empty_cache()
    tot = lambda f: sum(r["calls"] * f(r) for r in rows) / 1e3          # noqa: E731
    a, b_, ti, mix = tot(lambda r: r["auto"]), tot(lambda r: r["best"]), tot(lambda r: r["insitu"]), tot(lambda r: min(r["auto"], r["torch"]))
    print(f"GEMM ms/step (warm replay): auto {a:.3f}  best-per-launch {b_:.3f} ({100 * (a - b_) / a:.1f} % below auto)  "
          f"min(auto, torch) {mix:.3f} ({100 * (a - mix) / a:.1f} %)   in-step (cold operands, event-timed) {ti:.3f}")
    behind = [r for r in rows if r["torch"] < r["auto"] * 0.97]
    print(f"launches more than 3 % behind torch: {len(behind)}")
    for r in sorted(behind, key=lambda r: -(r["auto"] - r["torch"]) * r["calls"]):
        tm, tn, M, N, K, bt, akm, bkm, sp, f32, M2, N2, K2, bt2 = r["key"]
        print(f"  {M}x{N}x{K} x{bt} {'km' if akm else 'rm'}/{'km' if bkm else 'rm'}: auto {r['auto']:.1f} torch {r['torch']:.1f} us  x{r['calls']:.1f} per step "
              f"= {(r['auto'] - r['torch']) * r['calls'] / 1e3:.3f} ms/step")


if __name__ == "__main__":
    main()
