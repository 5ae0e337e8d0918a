#!/usr/bin/env python3
"""Benchmark of the hot path: VQA train samples/s (VL-T5-base, 36 regions, 20 question tokens, 5 answer tokens).

One "step" = VLT5VQA.train_step forward + backward + clip_grad_norm(5) + AdamW on one synthetic batch of 80 samples
per GPU (BASELINE.json configs[1]; batch 80 is what the reference's launch scripts use), dropout 0.1 ON, bf16 MFMA
compute with fp32 accumulation / fp32 master weights, inputs already resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0): metric/value (whole-job samples/s), roofline of the dominant kernel (HIP-event timed
inside this process) and the CPU baseline (the oracle restatement timed on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # this pool's driver only supports dmabuf IPC (RCCL across processes)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_BWD_GFLOP_PER_SAMPLE = 37.90          # 2*M*N*K per GEMM/bmm, L=20 V=36 T=5
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16


def gemm_schedule(cfg, B, L, V, T):
    """Every GEMM launch of one train step as (count, batch, M, N, K, a_kmajor, b_kmajor, out_f32): mirrors csrc/engine.hip
    (the weight-gradient GEMMs of all layers of a stack run as one batched launch per weight kind)."""
    d, inner, ff, Le, Ld, vocab, fd = cfg.d_model, cfg.num_heads * cfg.d_kv, cfg.d_ff, cfg.num_layers, cfg.num_decoder_layers, cfg.vocab_size, cfg.feat_dim
    S, Sx = L + V, L + V + 2
    M, Mx, Md = B * S, B * Sx, B * T
    sch = []

    def lin(layers, rows, n_out, k_in, dgrad_f32=True, batched_wgrad=True):
        sch.append((layers, 1, rows, n_out, k_in, 0, 0, 0))            # forward
        sch.append((layers, 1, rows, k_in, n_out, 0, 1, int(dgrad_f32)))   # dgrad
        if batched_wgrad:
            sch.append((1, layers, n_out, k_in, rows, 1, 1, 1))        # wgrad, grid.z = layers
        else:
            sch.append((layers, 1, n_out, k_in, rows, 1, 1, 1))
    lin(Le, M, 3 * inner, d)
    lin(Le, M, d, inner, False)
    lin(Le, M, ff, d)
    lin(Le, M, d, ff, False)
    lin(Ld, Md, 3 * inner, d)
    lin(Ld, Md, d, inner, False)
    lin(Ld, Md, inner, d)
    lin(Ld, Md, d, inner, False)
    lin(Ld, Md, ff, d)
    lin(Ld, Md, d, ff, False)
    lin(1, Mx, Ld * 2 * inner, d, True, False)
    lin(1, Md, vocab, d, True, False)
    sch.append((1, 1, B * V, d, fd, 0, 0, 1))
    sch.append((1, 1, d, fd, B * V, 1, 1, 1))
    return sch


def time_gemms(cfg, B, L, V, T, dev, reps=20):
    """HIP-event timing of each distinct GEMM launch of the step on the current stream (same tile / split-K / batching
    policy as the engine)."""
    from vqacl_amd import ops
    from vqacl_amd._lib import lib
    BF = torch.bfloat16
    rows = []
    slab = 8 * max(cfg.d_ff, 3 * cfg.num_heads * cfg.d_kv) * cfg.d_model * 4
    for count, batch, M, N, K, akm, bkm, of32 in gemm_schedule(cfg, B, L, V, T):
        A = torch.randn((batch, K, M) if akm else (batch, M, K), device=dev).to(BF)
        Bm = torch.randn((batch, K, N) if bkm else (batch, N, K), device=dev).to(BF)
        out = torch.empty(batch, M, N, device=dev, dtype=torch.float32 if of32 else BF)
        sk = lib().vlt5_gemm_auto_split(M, N, K, slab) if (of32 and bkm and batch == 1) else 1
        kw = dict(a_kmajor=bool(akm), b_kmajor=bool(bkm), out=out[0], split_k=sk, batch=batch,
                  batch_strides=(A.stride(0), Bm.stride(0), out.stride(0)))
        # descriptor built once, the loop only launches: a 400-row decoder GEMM is 6-8 us, a Python-side descriptor build is more
        import ctypes as C
        from vqacl_amd._lib import stream_ptr
        g, _, keep = ops.gemm_desc(A[0], Bm[0], M, N, K, **kw)
        fn, gp, sp = lib().vlt5_gemm_bf16, C.byref(g), stream_ptr()
        for _ in range(2):
            assert fn(gp, sp) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn(gp, sp)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        rows.append(dict(count=count, batch=batch, M=M, N=N, K=K, akm=akm, bkm=bkm, ms=ms, gflop=2.0 * batch * M * N * K / 1e9))
        del A, Bm, out
    return rows


def cpu_baseline(seconds_budget=25.0):
    """The oracle (CPU restatement of the reference path) timed on the host: fwd + bwd + clip + AdamW, dropout on."""
    from oracle import ref_cpu as R
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    cores = max(1, min(usable, 16))           # torch intra-op threads actually used: 16 was the fastest on the 256-core box
    torch.set_num_threads(cores)
    cfg = R.Cfg(dropout=0.1)
    Bc = 8
    model = R.OracleModel(cfg, seed=0)
    opt = R.HFAdamW(model.used, lr=1e-4, eps=1e-6, weight_decay=0.01)
    batch = R.synthetic_batch(cfg, B=Bc, L=20, V=36, T=5, seed=66666)
    times = []
    t_start = time.time()
    for it in range(3):
        t0 = time.time()
        model.zero_grad()
        out = model.train_step(batch, 0, 0.5, 0.3, training=True)
        out["loss"].backward()
        R.clip_grad_norm(list(model.used.values()), 5.0)
        opt.step()
        times.append(time.time() - t0)
        if time.time() - t_start > seconds_budget:
            break
    t = min(times[1:]) if len(times) > 1 else times[0]
    return dict(value=round(Bc / t, 3), unit="samples/s", cores=cores, kind="port",
                sample=f"{len(times)} train steps of batch {Bc} (L=20,V=36,T=5, dropout on, fp32 torch-CPU oracle), best of steps 2+")


def eager_gpu_baseline(dev, B=80, steps=5):
    """the restatement of the reference path run as plain PyTorch-ROCm eager ops on the same GPU (what a hipified
    reference would execute: torch ops + hipBLASLt GEMMs), fp32 like the reference and under bf16 autocast; same synthetic
    batch, dropout on, fwd + bwd + clip + AdamW.  A reported baseline beside `cpu_baseline`, never the product path."""
    from oracle import ref_cpu as R
    cfg = R.Cfg(dropout=0.1)
    params = {k: v.to(dev) for k, v in R.init_params(cfg, seed=0).items()}
    batch = {k: v.to(dev) for k, v in R.synthetic_batch(cfg, B=B, L=20, V=36, T=5, seed=66666).items()}
    out = {}
    torch.set_default_device(dev)                 # the restatement builds its index tensors with default-device factories
    try:
        for tag, amp in (("fp32", False), ("bf16_autocast", True)):
            model = R.OracleModel(cfg, params)
            opt = R.HFAdamW(model.used, lr=1e-4, eps=1e-6, weight_decay=0.01)

            def step():
                model.zero_grad()
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                    o = model.train_step(batch, 0, 0.5, 0.3, training=True)
                o["loss"].backward()
                R.clip_grad_norm(list(model.used.values()), 5.0)
                opt.step()
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            out[tag] = round(steps * B / (time.perf_counter() - t0), 1)
            del model, opt
    finally:
        torch.set_default_device("cpu")
    return dict(unit="samples/s", kind="torch-eager restatement on the same GPU", batch=B, **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager-baseline", action="store_true", help="also time the torch-eager restatement on the GPU (SURVEY 8d)")
    ap.add_argument("--overlap-optimizer", action="store_true", help="run the optimizer update on a second stream (see FusedAdamW)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("VQACL_FORCE_DIST") == "1"     # the env switch exercises the RCCL path on 1 GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        # high-priority RCCL stream: the bucket all-reduces must start when their gradients are ready, not queue behind the
        # backward GEMMs of the compute stream (measured with tools/dp_overlap_probe.py: at normal priority the casts of a
        # bucket released mid-backward only ran after backward had finished)
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, pg_options=opts)

    from oracle.ref_cpu import synthetic_batch, Cfg          # only the synthetic-input recipe and (rank 0) the cpu_baseline leg
    from vqacl_amd import VLT5VQA, VLT5Config, FusedAdamW, reference_param_groups
    cfg = VLT5Config(dropout_rate=0.1)
    torch.manual_seed(66666)
    model = VLT5VQA(cfg, device=dev)
    model.train()
    handle = model
    if distributed:
        from vqacl_amd.parallel import DataParallelVLT5
        handle = DataParallelVLT5(model)
    # --overlap-optimizer: the update of step n runs on a second stream under the forward of step n+1 (still inside the
    # timed region: the final synchronisation waits for every stream).  Measured 2 % SLOWER on MI355X (the 6.7 GB update
    # stream evicts the forward's operands from L2/MALL), so it is off by default.
    opt = FusedAdamW(reference_param_groups(model, 0.01), model, lr=1e-4, eps=1e-6, max_grad_norm=5.0,
                     overlap=args.overlap_optimizer)
    B, L, V, T = args.batch, 20, 36, 5
    batch = synthetic_batch(Cfg(), B=B, L=L, V=V, T=T, seed=66666 + rank, task_id=0)
    batch = {k: v.to(dev) for k, v in batch.items()}          # inputs resident in HBM before the timed region

    def step():
        res = handle.train_step(batch, 0, 0.5, 0.3)
        res["loss"].backward()
        opt.step()
        for p in model.parameters():
            p.grad = None
        return res["loss"]

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    final_loss = float(loss.detach())
    ms = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    out = {"metric": "vqa_train_samples_per_sec", "value": round(value, 2), "unit": "samples/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "VL-T5-base VQA v2 train step (fwd+bwd+clip+AdamW), 36x2048 regions, 20 question tokens, "
                                  "5 answer tokens, dropout 0.1", "batch_per_gpu": B, "global_batch": B * world,
                      "parallelism": f"dp{world}",
                      "grad_allreduce": (str(handle.grad_dtype).replace("torch.", "") if distributed else "none")},
           "samples_per_sec_per_gpu": round(value / world, 2), "final_loss": round(final_loss, 4),
           "step_tflops_per_gpu": round(FWD_BWD_GFLOP_PER_SAMPLE * B / ms, 2),
           "step_frac_of_mfma_peak": round(FWD_BWD_GFLOP_PER_SAMPLE * B / ms / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)}
    if rank == 0 and world == 1 and not distributed:
        # PCIe-inclusive rate (never `value`): the boundary normally hands over pinned HOST tensors (collate_fn output);
        # train_step then copies 23.6 MB of fp32 region features per batch of 80 before the engine starts.
        host = {k: v.cpu().pin_memory() for k, v in batch.items()}

        def step_host():
            res = handle.train_step(host, 0, 0.5, 0.3)
            res["loss"].backward()
            opt.step()
            for p in model.parameters():
                p.grad = None
        for _ in range(2):
            step_host()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            step_host()
        torch.cuda.synchronize()
        out["samples_per_sec_pcie_inclusive"] = round(5 * B / (time.perf_counter() - t1), 2)
        # f-2 feed: the same step fed from the HBM-resident bf16 feature store (vqacl_amd/feed.py): a fresh random draw of
        # 80 images out of 4096 per step, only ids / labels / slot indices come from the host.  Also never `value`.
        from vqacl_amd.feed import FeatureStore
        n_img = 4096
        store = FeatureStore(n_img, n_boxes=V, feat_dim=cfg.feat_dim, device=dev)
        for a in range(0, n_img, 256):
            store.put(list(range(a, a + 256)), torch.relu(torch.randn(256, V, cfg.feat_dim, device=dev)) * 1.5,
                      torch.rand(256, V, 4, device=dev).sort(-1).values)
        small = {k: v for k, v in host.items() if k not in ("vis_feats", "boxes")}
        draws = [torch.randint(0, n_img, (B,)).tolist() for _ in range(8)]

        def step_store(i):
            fed = dict(small)
            fed["feat_ref"] = store.ref(draws[i % 8])
            res = handle.train_step(fed, 0, 0.5, 0.3)
            res["loss"].backward()
            opt.step()
            for p in model.parameters():
                p.grad = None
        for i in range(2):
            step_store(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(8):
            step_store(i)
        torch.cuda.synchronize()
        out["samples_per_sec_store_feed"] = round(8 * B / (time.perf_counter() - t1), 2)
        # the gather kernel against the HBM roofline: algorithmic bytes = rows read + rows written (bf16 features + f32 boxes)
        slots = store.slots(draws[0])
        from vqacl_amd._lib import lib, ptr, stream_ptr
        of = torch.empty(B, V, cfg.feat_dim, dtype=torch.bfloat16, device=dev)
        ob = torch.empty(B, V, 4, device=dev)
        gargs = (ptr(store.feats), ptr(store.boxes), ptr(slots), store.capacity, ptr(of), ptr(ob), B, V, cfg.feat_dim, stream_ptr())
        fn = lib().vlt5_feat_gather                      # pre-bound arguments: the launch loop must not be host-bound (a launch is ~6 us)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(10):
            fn(*gargs)
        e0.record()
        for _ in range(200):
            fn(*gargs)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        gbytes = 2 * B * V * (cfg.feat_dim * 2 + 16) / 1e9
        out["feed"] = {"kernel": "feat_gather_kernel", "bound": "hbm", "bytes_per_launch": int(gbytes * 1e9), "us": round(us, 2),
                       "achieved": round(gbytes / (us * 1e-6), 1), "peak": 8000.0, "unit": "GB/s",
                       "frac": round(gbytes / (us * 1e-6) / 8000.0, 4), "store_images": n_img,
                       "store_gb": round(n_img * V * (cfg.feat_dim * 2 + 16) / 1e9, 3)}
        del store
    if rank == 0 and world == 1 and not distributed and not args.no_roofline:
        rows = time_gemms(cfg, B, L, V, T, dev)
        launches = sum(r["count"] for r in rows)
        tot_ms = sum(r["count"] * r["ms"] for r in rows)
        tot_gflop = sum(r["count"] * r["gflop"] for r in rows)
        achieved = tot_gflop / tot_ms                      # GFLOP/ms == TFLOP/s
        out["roofline"] = {"bound": "mfma", "kernel": "gemm_kernel<BM,BN,AKM,BKM> (all instantiations)",
                           "achieved": round(achieved, 2), "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(achieved / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "traffic": None,
                           "launches_per_step": launches, "avg_launch_us": round(tot_ms / launches * 1e3, 2),
                           "gflop_per_launch": round(tot_gflop / launches, 3), "gemm_ms_per_step": round(tot_ms, 3)}
        # HBM traffic of the same kernel family from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
        # separate runs, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md): launch-weighted bytes per launch
        import glob
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.json")))
        if pmcs:
            g = [r for r in json.load(open(pmcs[-1])) if "gemm_kernel" in r["kernel"]]
            if g:
                out["roofline"]["traffic"] = round(sum(r["calls"] * r["hbm_mb"] for r in g) / sum(r["calls"] for r in g) * 1e6)
                out["roofline"]["traffic_unit"] = f"bytes/launch (PMC, profiles/{os.path.basename(pmcs[-1])})"
                out["roofline"]["algorithmic_bytes_per_launch"] = round(sum(
                    r["count"] * r["batch"] * 2 * (r["M"] * r["K"] + r["N"] * r["K"] + r["M"] * r["N"] * (2 if r["akm"] else 1)) for r in rows) / launches)
        worst = sorted(rows, key=lambda r: -r["count"] * r["ms"])[:6]
        out["roofline"]["top_shapes"] = [dict(M=r["M"], N=r["N"], K=r["K"], akm=r["akm"], bkm=r["bkm"], count=r["count"], batch=r["batch"],
                                              us=round(r["ms"] * 1e3, 1), tflops=round(r["gflop"] / r["ms"], 1)) for r in worst]
    if rank == 0 and world == 1 and not distributed and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0 and world == 1 and not distributed and args.eager_baseline:
        del model, opt
        torch.cuda.empty_cache()
        out["eager_gpu_baseline"] = eager_gpu_baseline(dev, B)
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
