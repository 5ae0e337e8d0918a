#!/usr/bin/env python3
"""Benchmark of the hot path: VQA train samples/s (VL-T5-base, 36 regions, 20 question tokens, 5 answer tokens).

One "step" = VLT5VQA.train_step forward + backward + clip_grad_norm(5) + AdamW on one synthetic batch of 80 samples
per GPU (BASELINE.json configs[1]; batch 80 is what the reference's launch scripts use), dropout 0.1 ON, bf16 MFMA
compute with fp32 accumulation / fp32 master weights.

The timed loop is the designed feed path: the region features of 4096 synthetic images are resident in HBM
(bf16 feature store) before the timed region starts, and EVERY step draws a fresh random batch of 80 of them -- only token ids,
labels, scores and the 80 slot indices come from the host, as the reference's collate_fn would hand them over.  Side values
(never `value`): the same step re-fed from one device-resident f32 batch, and the PCIe-inclusive step (pinned host f32 batch).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0): metric/value (whole-job samples/s), `roofline` of the dominant kernel family (every GEMM
dispatch of real train steps timed in situ with HIP events attached to the dispatch), `parity` of the benched configuration
against the CPU oracle, and `cpu_baseline` (the oracle timed on the host cores, bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # this pool's driver only supports dmabuf IPC (RCCL across processes)


def arm_stall_dump(role):
    """SIGUSR1 -> the Python stacks of every thread of THIS process, written to $VQACL_STALL_DUMP_DIR/<role>_pid<pid>.txt (stderr when the
    variable is unset).  The launcher sends it to its ranks before it gives up on them, tests/test_gpu_bench_line.py sends it to a launch
    that ran into its bound: a stalled multi-rank launch leaves the call every rank sits in instead of a bare timeout."""
    import faulthandler
    import signal
    target = sys.stderr
    d = os.environ.get("VQACL_STALL_DUMP_DIR")
    if d:
        try:
            os.makedirs(d, exist_ok=True)
            target = open(os.path.join(d, f"{role}_pid{os.getpid()}.txt"), "w")
        except OSError:
            target = sys.stderr
    faulthandler.register(signal.SIGUSR1, file=target, all_threads=True, chain=False)
    return target


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=80)
    ap.add_argument("--store-images", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-side-values", action="store_true", help="skip the resident-batch / PCIe / feed-kernel / warm-replay side figures")
    ap.add_argument("--eager-baseline", action="store_true", help="also time the torch-eager restatement on the GPU (SURVEY 8d)")
    ap.add_argument("--overlap-optimizer", action="store_true", help="run the optimizer update on a second stream (see FusedAdamW)")
    ap.add_argument("--dp-algo", default=os.environ.get("VQACL_DP_ALGO", "auto"), help="gradient exchange of the DP wrapper (parallel.py)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run through the rank launcher and the data-parallel wrapper (RCCL) even with --gpus 1")
    ap.add_argument("--rehearsal", action="store_true",
                    help="dress rehearsal of the N-rank path on ONE GPU: the launcher starts --gpus N rank processes that all use cuda:0 "
                         "and exchange gradients through gloo (RCCL cannot place two ranks on one device); the line is flagged "
                         "\"rehearsal\": true and carries NO value -- it proves launcher, rendezvous, wrapper, events and JSON relay, not speed")
    ap.add_argument("--dry-launch", action="store_true", help="print the environment and command of every rank the launcher would start, start nothing")
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="seconds before the launcher gives up on its ranks")
    return ap.parse_args(argv)


# ---- rank launcher ----------------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` (no WORLD_SIZE in the environment) has to BE an N-rank job: this parent process starts N copies of
# itself, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what `python -m torch.distributed.run` would export),
# relays rank 0's JSON line as its own last stdout line and fails if any rank fails.  It runs before torch is imported and never
# touches the GPU (a process that initialised HIP must not exec or fork workers on this pool).  Under torch.distributed.run the
# environment already carries WORLD_SIZE: then this process is a rank and --gpus must agree with it.
RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "VQACL_FORCE_DIST")


def launch_plan(args, argv, environ=None, port=None):
    """None when this process is itself a rank (or a plain single-process run); otherwise the list of (env additions, command)
    of the ranks to start.  Raises SystemExit when --gpus contradicts an inherited WORLD_SIZE."""
    environ = os.environ if environ is None else environ
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    inherited = environ.get("WORLD_SIZE")
    if inherited is not None:
        if int(inherited) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={inherited} in the environment: the launcher that set "
                             f"WORLD_SIZE started {inherited} ranks; pass --gpus {inherited} or unset WORLD_SIZE")
        return None                                        # a rank of somebody else's launcher (torch.distributed.run)
    if args.gpus == 1 and not (args.force_dist or args.rehearsal or environ.get("VQACL_FORCE_DIST") == "1"):
        return None                                        # the plain single-process run
    if port is None:
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    child_argv = [a for a in argv if a not in ("--dry-launch",)]
    plan = []
    for r in range(args.gpus):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
        if args.gpus == 1:
            env["VQACL_FORCE_DIST"] = "1"
        plan.append((env, [sys.executable, os.path.abspath(__file__)] + child_argv))
    return plan


def profiler_preload(environ=None):
    """The profiler tool library preloaded into this process, if any (rocprofv3 sets LD_PRELOAD / ROCP_TOOL_LIBRARIES /
    HSA_TOOLS_LIB to its librocprofiler-sdk-tool); None otherwise."""
    environ = os.environ if environ is None else environ
    for var in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD"):
        val = environ.get(var, "")
        if "rocprof" in val.lower() or "roctracer" in val.lower():
            return f"{var}={val}"
    return None


def run_launcher(args, argv):
    """Start the ranks of launch_plan(), wait, relay.  Returns the exit code of the job."""
    import subprocess
    import threading
    plan = launch_plan(args, argv)
    if plan is None:
        return None
    if args.dry_launch:
        for env, cmd in plan:
            print(json.dumps({"env": env, "cmd": cmd}), flush=True)
        return 0
    tool = profiler_preload()
    if tool:
        # a profiler's preloaded library has initialised the GPU in THIS process already: starting rank processes from it is the
        # forbidden "program started from a GPU-initialised process" on this pool -- the profiled process has to BE the rank
        print(f"bench.py launcher: refusing to start ranks under a preloaded profiler ({tool}).  Run the rank itself under the "
              f"profiler: set WORLD_SIZE={args.gpus} RANK=<r> LOCAL_RANK=<r> MASTER_ADDR=127.0.0.1 MASTER_PORT=<port>"
              + (" VQACL_FORCE_DIST=1" if args.gpus == 1 else "") + " in the environment (tools/profile_round.sh does) and put "
              "`python3 bench.py ...` after `--`", file=sys.stderr, flush=True)
        return 2
    procs, lines = [], []
    arm_stall_dump("launcher")

    def pump(rank, stream):            # every rank's stdout: rank 0's last JSON object is the bench line, the rest goes to stderr
        for raw in stream:
            line = raw.rstrip("\n")
            if rank == 0 and line.startswith("{") and '"metric"' in line:
                lines.append(line)
            else:
                print(f"[rank {rank}] {line}", file=sys.stderr, flush=True)
    threads = []
    for r, (env, cmd) in enumerate(plan):
        p = subprocess.Popen(cmd, env={**os.environ, **env}, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1)
        procs.append(p)
        t = threading.Thread(target=pump, args=(r, p.stdout), daemon=True)
        t.start()
        threads.append(t)
    deadline = time.time() + args.launch_timeout
    rc, alive = 0, set(range(len(procs)))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is not None:
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py launcher: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
        if (rc != 0 or time.time() > deadline) and alive:
            if rc == 0:
                rc = 124
                print("bench.py launcher: timed out waiting for the ranks; asking them for their stacks (SIGUSR1)", file=sys.stderr, flush=True)
                import signal
                for r in alive:                            # (every rank armed arm_stall_dump() before it imported torch)
                    procs[r].send_signal(signal.SIGUSR1)
                time.sleep(3.0)
            for r in alive:                                # exactly the processes started above
                procs[r].terminate()
            for r in alive:
                try:
                    procs[r].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            alive.clear()
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=5)
    if rc == 0 and not lines:
        print("bench.py launcher: rank 0 printed no JSON line", file=sys.stderr, flush=True)
        rc = 1
    if lines:
        sys.stderr.flush()
        print(lines[-1], flush=True)
    return rc


if __name__ == "__main__":
    _rc = run_launcher(parse_args(), sys.argv[1:])
    if _rc is not None:
        sys.exit(_rc)

if __name__ == "__main__":
    arm_stall_dump("rank%s" % os.environ.get("RANK", "0"))

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_BWD_GFLOP_PER_SAMPLE = 37.90          # 2*M*N*K per GEMM/bmm, L=20 V=36 T=5
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16
N_QUES, N_CATE, VOCAB, FEAT_DIM = 10, 80, 32200, 2048


def synthetic_batch(B, L=20, V=36, T=5, seed=66666, task_id=0, cate_group=0, with_feats=True):
    """Synthetic batch of SURVEY 8(d) in the reference's collate_fn schema: non-negative sparse-ish region features
    (relu(N(0,1)) * 1.5), sorted box corners in [0,1], ragged questions (pad 0, row 0 forced to the full length so S = L + V),
    ragged answers ending in EOS = 1 (pad -100), one-hot task / category labels, answer scores out of {0.3, 0.6, 0.9, 1.0}."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    if with_feats:
        out["vis_feats"] = torch.relu(torch.randn(B, V, FEAT_DIM, generator=g)) * 1.5
        xs = torch.rand(B, V, 2, generator=g).sort(dim=2).values
        ys = torch.rand(B, V, 2, generator=g).sort(dim=2).values
        out["boxes"] = torch.stack([xs[..., 0], ys[..., 0], xs[..., 1], ys[..., 1]], dim=2)
    ids = torch.randint(2, 32000, (B, L), generator=g)
    lens = torch.randint(min(6, L), L + 1, (B,), generator=g)
    lens[0] = L
    out["input_ids"] = ids * (torch.arange(L)[None, :] < lens[:, None])
    tgt = torch.randint(2, 32000, (B, T), generator=g)
    tl = torch.randint(2, T + 1, (B,), generator=g)
    tl[0] = T
    pos = torch.arange(T)[None, :]
    tgt = torch.where(pos == (tl[:, None] - 1), torch.ones_like(tgt), tgt)
    out["target_ids"] = torch.where(pos < tl[:, None], tgt, torch.full_like(tgt, -100))
    out["ques_labels"] = torch.zeros(B, N_QUES)
    out["ques_labels"][:, task_id] = 1
    cate_ids = cate_group * 16 + torch.randint(0, 16, (B,), generator=g)
    out["cate_labels"] = torch.zeros(B, N_CATE).scatter_(1, cate_ids[:, None] % N_CATE, 1.0)
    out["scores"] = torch.tensor([0.3, 0.6, 0.9, 1.0])[torch.randint(0, 4, (B,), generator=g)]
    return out


def decode_weight_bytes(cfg):
    """bf16 weight bytes one greedy token-step reads: per decoder layer the self-attention q|k|v and o, the cross-attention q and o
    (cross k|v are projected once per batch, not per token), wi and wo; plus the tied embedding as the lm_head."""
    d, inner, ff = cfg.d_model, cfg.num_heads * cfg.d_kv, cfg.d_ff
    ffw = (3 if cfg.is_gated_act else 2) * d * ff
    return int(2 * (cfg.num_decoder_layers * (3 * inner * d + 3 * inner * d + ffw) + cfg.vocab_size * d))


def gemm_schedule(cfg, B, L, V, T):
    """Every GEMM launch of one train step as (count, batch, M, N, K, a_kmajor, b_kmajor, out_f32): mirrors csrc/engine.hip
    (the weight-gradient GEMMs of all layers of a stack run as one batched launch per weight kind).  Only used by the
    warm-replay side figure (`frac_warm`)."""
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


def time_gemms_warm(cfg, B, L, V, T, dev, reps=20):
    """Warm replay: each distinct GEMM shape of the step launched `reps` times back to back on the same operands (L2/MALL warm).
    An upper bound of what the kernels do in situ; reported as `frac_warm`, never as `frac`."""
    from vqacl_amd import ops
    from vqacl_amd._lib import lib, stream_ptr
    BF = torch.bfloat16
    tot_ms = tot_gflop = 0.0
    slab = 8 * max(cfg.d_ff, 3 * cfg.num_heads * cfg.d_kv) * cfg.d_model * 4
    for count, batch, M, N, K, akm, bkm, of32 in gemm_schedule(cfg, B, L, V, T):
        A = torch.randn((batch, K, M) if akm else (batch, M, K), device=dev).to(BF)
        Bm = torch.randn((batch, K, N) if bkm else (batch, N, K), device=dev).to(BF)
        out = torch.empty(batch, M, N, device=dev, dtype=torch.float32 if of32 else BF)
        sk = lib().vlt5_gemm_auto_split(M, N, K, slab) if (of32 and bkm and batch == 1) else 1
        g, _, keep = ops.gemm_desc(A[0], Bm[0], M, N, K, a_kmajor=bool(akm), b_kmajor=bool(bkm), out=out[0], split_k=sk, batch=batch,
                                   batch_strides=(A.stride(0), Bm.stride(0), out.stride(0)))
        fn, gp, sp = lib().vlt5_gemm_bf16, C.byref(g), stream_ptr()
        for _ in range(2):
            assert fn(gp, sp) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn(gp, sp)
        e1.record()
        e1.synchronize()
        tot_ms += count * e0.elapsed_time(e1) / reps
        tot_gflop += count * 2.0 * batch * M * N * K / 1e9
        del A, Bm, out
    return tot_gflop / tot_ms, tot_ms


def insitu_gemm_roofline(step_fn, n_steps):
    """Every gemm_kernel dispatch of `n_steps` REAL train steps (fresh store-fed batches, dropout on, optimizer included) timed
    with HIP events attached to the dispatch (vlt5_gemm_timing_*): achieved = sum of 2*M*N*K*batch / sum of the kernels' durations."""
    from vqacl_amd._lib import GemmTimingRec, lib
    cap = 1024 * n_steps
    assert lib().vlt5_gemm_timing_enable(cap) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_steps):
        step_fn(i)
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) / n_steps * 1e3
    recs = (GemmTimingRec * cap)()
    n = lib().vlt5_gemm_timing_collect(recs, cap)
    lib().vlt5_gemm_timing_enable(0)
    assert 0 < n <= cap, n
    by = {}
    tot_ms = tot_gflop = alg_bytes = 0.0
    for r in recs[:n]:
        gf = 2.0 * (r.batch * r.M * r.N * r.K + r.batch2 * r.M2 * r.N2 * r.K2) / 1e9          # (a grouped launch carries a second problem)
        tot_ms += r.ms
        tot_gflop += gf
        # algorithmic bytes of a launch: both bf16 operands once + the output once (f32 or bf16; split-K slabs count as written)
        alg_bytes += r.batch * (2.0 * (r.M * r.K + r.N * r.K) + (4 if r.out_f32 else 2) * r.M * r.N * max(r.splits, 1))
        alg_bytes += r.batch2 * (2.0 * (r.M2 * r.K2 + r.N2 * r.K2) + 4 * r.M2 * r.N2) if r.M2 else 0.0
        if (r.tile_m, r.tile_n) == (128, 384):      # the fused q|k|v projection + attention core (its 2*M*N*K is the projection alone)
            key = "qkv_attn_fwd_kernel<128,384> (projection + attention core)"
        else:
            key = (f"gemm_kernel<{r.tile_m},{r.tile_n},{'km' if r.a_kmajor else 'rm'},{'km' if r.b_kmajor else 'rm'}>")
        k = by.setdefault(key, [0, 0.0, 0.0])
        k[0] += 1
        k[1] += r.ms
        k[2] += gf
    launches = n / n_steps
    achieved = tot_gflop / tot_ms                       # GFLOP/ms == TFLOP/s
    per_kernel = {k: dict(calls_per_step=round(v[0] / n_steps, 1), avg_us=round(v[1] / v[0] * 1e3, 2), ms_per_step=round(v[1] / n_steps, 3),
                          tflops=round(v[2] / v[1], 1)) for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])}
    return dict(bound="mfma", kernel="gemm_kernel<BM,BN,AKM,BKM> (all instantiations) + qkv_attn_fwd_kernel, in situ", achieved=round(achieved, 2),
                peak=MFMA_BF16_DENSE_PEAK_TFLOPS, unit="TFLOP/s", frac=round(achieved / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), traffic=None,
                launches_per_step=round(launches, 1), avg_launch_us=round(tot_ms / n * 1e3, 2),
                gflop_per_launch=round(tot_gflop / n, 3), gemm_ms_per_step=round(tot_ms / n_steps, 3),
                algorithmic_bytes_per_launch=round(alg_bytes / n),
                timed_steps=n_steps, ms_per_step_while_timed=round(wall_ms, 3), per_kernel=per_kernel)


def parity_vs_oracle(model, dev, B):
    """Parity figure of the BENCHED configuration (VL-T5-base, B = 80 launch shapes, current weights) against the fp32 CPU oracle:
    forward of the whole batch with dropout off, compared on 4 of the 80 samples (per-sample independence of the path is proven
    bit-exact by tests/test_gpu_model.py::test_full_size_batch_properties); gradients from a B = 4 step of the same weights."""
    from oracle import ref_cpu as R
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
    ocfg = R.Cfg(dropout=0.0)
    batch = synthetic_batch(B, seed=424242)
    pick = [0, 1, B // 2, B - 1]
    sub = {k: v[pick] for k, v in batch.items()}
    params = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items() if k in R.param_shapes(ocfg)}
    Qp, Vp = model.Q_prototype.detach().cpu().clone(), model.V_prototype.detach().cpu().clone()
    was_training, p_drop = model.training, model.cfg.dropout_rate
    model.cfg.dropout_rate = 0.0
    model.train()
    try:
        out = model(input_ids=batch["input_ids"], vis_inputs=(batch["vis_feats"], batch["boxes"]), labels=batch["target_ids"],
                    proto_update=False)
        logits = out["logits"][pick].float().cpu()
        loss_tok = out["loss"].detach().view(B, -1)[pick].float().cpu()
        idx = (out["max_idx_Q"].cpu()[pick], out["max_idx_V"].cpu()[pick])
        oracle = R.OracleModel(ocfg, params)
        oracle.state.Q_prototype, oracle.state.V_prototype = Qp.clone(), Vp.clone()
        o = R.vlt5_forward(oracle.P, oracle.state, ocfg, input_ids=sub["input_ids"], vis_feats=sub["vis_feats"], boxes=sub["boxes"],
                           labels=sub["target_ids"], proto_update=False, training=False)
        ol = o["logits"].detach()
        res = dict(samples=pick, logits_rel_max_err=float((logits - ol).abs().max() / ol.abs().max()),
                   loss_tok_abs_err=float((loss_tok.flatten() - o["loss"].detach()).abs().max()))
        # integer outputs: retrieved prototype indices, exact wherever the oracle's top-2 cosine margin exceeds 1e-2
        h = o["encoder_hidden_states"].detach()
        exact, gated = 0, 0
        for protos, pooled, mine, ref in ((Qp, h[:, :20].mean(1), idx[0], o["max_idx_Q"]), (Vp, h[:, 20:].mean(1), idx[1], o["max_idx_V"])):
            a = torch.nn.functional.normalize(torch.tanh(protos), dim=1)
            b = torch.nn.functional.normalize(torch.tanh(pooled), dim=1)
            top = (b @ a.t()).topk(2, dim=1).values
            ok = (top[:, 0] - top[:, 1]) > 1e-2
            gated += int(ok.sum())
            exact += int((mine[ok] == ref[ok]).sum())
        res["proto_idx_checked"], res["proto_idx_equal"] = gated, exact
        # gradients: the same 4 samples as a B = 4 step (fused train_step reduction, dropout off, prototypes retrieved but not
        # updated so that both sides see the same state) against the oracle's backward of the forward above
        for p in model.parameters():
            p.grad = None
        r = model(input_ids=sub["input_ids"], vis_inputs=(sub["vis_feats"], sub["boxes"]), labels=sub["target_ids"],
                  proto_update=False, scores=sub["scores"])
        r["loss_reduced"].backward()
        lo = R.train_step_loss(o["loss"], sub["target_ids"], sub["scores"])
        lo.backward()
        res["loss_abs_err"] = abs(float(r["loss_reduced"].detach()) - float(lo.detach()))
        worst, worst_name = 1.0, None
        named = dict(model.named_parameters())
        for k, p in oracle.P.items():
            if p.grad is None or k not in named or named[k].grad is None or float(p.grad.abs().max()) < 1e-10:
                continue
            a, b = named[k].grad.float().cpu().flatten(), p.grad.flatten()
            c = float(torch.dot(a, b) / (a.norm() * b.norm()).clamp(min=1e-30))
            if c < worst:
                worst, worst_name = c, k
        res["worst_grad_cos"], res["worst_grad_tensor"] = round(worst, 5), worst_name
        res["logits_rel_max_err"] = round(res["logits_rel_max_err"], 5)
        res["loss_tok_abs_err"] = round(res["loss_tok_abs_err"], 5)
        res["loss_abs_err"] = round(res["loss_abs_err"], 6)
        res["tolerance"] = "logits 1e-2 rel, per-token loss 2e-2 abs, reduced loss 1e-2 abs, grad cos >= 0.99, indices exact where the top-2 margin > 1e-2 (tests/test_gpu_model.py::test_benched_shape_b80_against_the_oracle)"
        return res
    finally:
        for p in model.parameters():
            p.grad = None
        model.cfg.dropout_rate = p_drop
        model.train(was_training)


def cpu_baseline(budget_s=30.0):
    """The oracle (CPU restatement of the reference path) timed on the host, SURVEY 8(d) protocol: fwd + bwd + clip + AdamW, dropout
    on, fp32 torch-CPU ops, at the two batch sizes the survey names -- B = 4 (BASELINE configs[0], the reference's own CPU-runnable
    case) and B = 80 (the benched configuration) -- warm-up steps first, then the MEDIAN of the timed steps.  The step counts are
    what a bounded sample allows (the survey's 3 + 10 steps at B = 80 would be a minute of CPU work): stated in `sample`."""
    from oracle import ref_cpu as R
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    cores = max(1, min(usable, 16))           # torch intra-op threads actually used: 16 was the fastest on the 256-core box
    torch.set_num_threads(cores)
    cfg = R.Cfg(dropout=0.1)
    res, t_all = {}, time.time()
    for Bc, warm, timed in ((4, 2, 7), (80, 1, 3)):
        model = R.OracleModel(cfg, seed=0)
        opt = R.HFAdamW(model.used, lr=1e-4, eps=1e-6, weight_decay=0.01)
        batch = synthetic_batch(Bc, seed=66666)
        times = []
        for i in range(warm + timed):
            t0 = time.time()
            model.zero_grad()
            out = model.train_step(batch, 0, 0.5, 0.3, training=True)
            out["loss"].backward()
            R.clip_grad_norm(list(model.used.values()), 5.0)
            opt.step()
            if i >= warm:
                times.append(time.time() - t0)
            if time.time() - t_all > budget_s and len(times) >= 1:
                break                                   # (a slow host: keep the sample bounded, report what was timed)
        times.sort()
        res[Bc] = (Bc / times[len(times) // 2], len(times), warm)
        del model, opt
    return dict(value=round(res[80][0], 3), unit="samples/s", cores=cores, kind="port", value_b4=round(res[4][0], 3),
                sample=(f"median step of the fp32 torch-CPU oracle (L=20,V=36,T=5, dropout on, fwd+bwd+clip+AdamW): B=80 "
                        f"{res[80][1]} timed steps after {res[80][2]} warm-up -> value; B=4 {res[4][1]} timed after {res[4][2]} warm-up -> value_b4; "
                        f"{time.time() - t_all:.0f} s of CPU work in all"))


def committed_profile(pattern):
    """The newest committed profile file matching `pattern` (profiles/rNN_x_<pattern>), rounds newest first."""
    import glob
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"{rnd}*_{pattern}")))
        if files:
            return files[-1]
    return None


def trace_roofline(gflop_per_step):
    """`frac` recomputed from the committed rocprofv3 kernel trace (profiles/rNN*_kernel_stats_bench_b80.txt, the same bench command
    under --kernel-trace): time of the GEMM family per optimizer step from the trace, FLOPs per step from the in-situ records of this
    run (same launches).  The event-timed `frac` perturbs the step it measures; this one does not.  Only quoted when the trace was
    taken with THIS build: the stats file carries the hash of the kernel sources (tools/rocpd_stats.py), a trace of another build is
    named with `trace_stale` and no figure."""
    import re
    from vqacl_amd.build import source_hash
    f = committed_profile("kernel_stats_bench_b80.txt")
    if not f:
        return None
    steps, fam_ms, sha, col = None, 0.0, None, None
    for line in open(f):
        m = re.match(r"# (\d+) optimizer steps in the trace", line)
        if m:
            steps = int(m.group(1))
        m = re.match(r"# source_sha16 (\w+)", line)
        if m:
            sha = m.group(1)
        if line.startswith("kernel "):
            col = line.split().index("total_ms") - len(line.split())        # column of total_ms counted from the end of the row
            continue
        if line.startswith("#") or col is None:
            continue
        if any(k in line for k in ("gemm_kernel<", "qkv_attn_fwd", "skinny_kernel<", "dec_attn_fwd_kernel")):
            fam_ms += float(line.split()[col])
    src = f"profiles/{os.path.basename(f)}"
    if sha != source_hash():
        return dict(trace_stale=True, trace_source=src, trace_note="kernel trace of another build (source hash differs): no frac_trace quoted")
    if not steps or fam_ms <= 0:
        return None
    ms = fam_ms / steps
    return dict(frac_trace=round(gflop_per_step / ms / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), gemm_ms_per_step_trace=round(ms, 3), trace_source=src)


def eager_gpu_baseline(dev, B=80, steps=5):
    """the restatement of the reference path run as plain PyTorch-ROCm eager ops on the same GPU (what a hipified
    reference would execute: torch ops + hipBLASLt GEMMs), fp32 like the reference and under bf16 autocast; same synthetic
    batch, dropout on, fwd + bwd + clip + AdamW.  A reported baseline beside `cpu_baseline`, never the product path."""
    from oracle import ref_cpu as R
    cfg = R.Cfg(dropout=0.1)
    params = {k: v.to(dev) for k, v in R.init_params(cfg, seed=0).items()}
    batch = {k: v.to(dev) for k, v in synthetic_batch(B, seed=66666).items()}
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


# forward + backward GFLOP per sample (2*M*N*K per GEMM / bmm) of the BASELINE shapes
OTHER_CONFIGS = [
    # key, what, config overrides, B, L, V, T, GFLOP per sample
    ("c1_b4", "VL-T5-base, B = 4 (BASELINE configs[0] shape)", {}, 4, 20, 36, 5, 37.90),
    ("c4_v16", "VL-T5-base NExT-QA, 16 frames, L = 23, T = 6, B = 80 (configs[3])", dict(n_ques=8), 80, 23, 16, 6, 28.2),
    ("c4_v32", "VL-T5-base NExT-QA, 32 frames, L = 23, T = 6, B = 80 (configs[3])", dict(n_ques=8), 80, 23, 32, 6, 36.2),
    ("c5_large_b32", "VL-T5-large, B = 32 (configs[4])", dict(d_model=1024, num_heads=16, d_ff=4096, num_layers=24), 32, 20, 36, 5, 132.0),
    ("c5_large_b80", "VL-T5-large, B = 80 per GPU (288 GB holds it)", dict(d_model=1024, num_heads=16, d_ff=4096, num_layers=24), 80, 20, 36, 5, 132.0),
]


def other_config_lines(dev, steps=5, warm=3):
    """Side values (never `value`): the train step of the other BASELINE shapes -- 5 timed steps each after 3 warm-up steps on one
    device-resident synthetic batch (dropout 0.1, fwd + bwd + clip + AdamW); ms/step, samples/s, the step as a fraction of the dense
    bf16 MFMA peak, and the optimizer's share (event-timed in the same steps; for the large model also the update of ONE EIGHTH of the
    parameters -- what a rank of the ZeRO-1 sharded step at N = 8 runs)."""
    from vqacl_amd import VLT5VQA, VLT5Config, FusedAdamW, reference_param_groups
    from vqacl_amd import _lib as L
    from vqacl_amd._lib import lib
    out = {}
    for key, what, kw, B, Lq, V, T, gflop in OTHER_CONFIGS:
        cfg = VLT5Config(dropout_rate=0.1, **kw)
        torch.manual_seed(1)
        model = VLT5VQA(cfg, device=dev)
        model.train()
        opt = FusedAdamW(reference_param_groups(model, 0.01), model, lr=1e-4, eps=1e-6, max_grad_norm=5.0)
        batch = synthetic_batch(B, Lq, V, T, seed=3)
        if cfg.n_ques != N_QUES:
            batch["ques_labels"] = batch["ques_labels"][:, :cfg.n_ques].contiguous()
        batch = {k: v.to(dev) for k, v in batch.items()}
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

        def step(i=None):
            model.train_step(batch, 0, 0.5, 0.3)["loss"].backward()
            if i is not None:
                ev[i][0].record()
            opt.step()
            if i is not None:
                ev[i][1].record()
            for p in model.parameters():
                p.grad = None
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        adam = sorted(a.elapsed_time(b) for a, b in ev)[steps // 2]
        n_par = int(opt._used_end)
        rec = {"what": what, "batch": B, "ms_per_step": round(ms, 3), "samples_per_sec": round(B * 1e3 / ms, 1),
               "gflop_per_sample": gflop, "step_tflops": round(gflop * B / ms, 1),
               "step_frac_of_mfma_peak": round(gflop * B / ms / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
               "params_m": round(n_par / 1e6, 1), "clip_adamw_ms": round(adam, 3), "clip_adamw_share": round(adam / ms, 3)}
        if key.startswith("c5"):
            # the update of one eighth of the flat buffers (lr = 0: the weights stay as they are): the optimizer pass of a ZeRO-1 rank at N = 8
            n8 = n_par // 8 // 64 * 64
            flat, grad, bf = model._flat, model._flat_grad, model._flat_bf16
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            args = (L.vp(flat.data_ptr()), L.vp(grad.data_ptr()), L.vp(opt._m.data_ptr()), L.vp(opt._v.data_ptr()), L.vp(bf.data_ptr()), n8,
                    0.0, 0.9, 0.999, 1e-6, 0.0, 5, L.vp(opt._total_sq.data_ptr()), 5.0, 1)
            for rep in range(3):
                if rep == 2:
                    a0.record()
                assert lib().vlt5_adamw_step(*args, L.stream_ptr()) == 0
            a1.record()
            a1.synchronize()
            rec["adamw_ms_one_eighth_shard"] = round(a0.elapsed_time(a1), 3)
            rec["ms_per_step_with_sharded_update_estimate"] = round(ms - adam + a0.elapsed_time(a1), 3)
        out[key] = rec
        del model, opt, batch
        torch.cuda.empty_cache()
    return out


def feed_figures(store, slots_list, cfg, B, V, dev, trace_us=None):
    """feat_gather_kernel against the HBM roofline.  `us` (-> frac): a train of launches that each gather a DIFFERENT random batch out
    of the 0.6 GB store into rotating output buffers (between two reads of an image lie several hundred MB of other rows: the store is
    larger than the 256 MB Infinity Cache, so the rows come from HBM as in a train step), one event pair around the whole train;
    `us_replay_cache_resident`: 200 back-to-back gathers of the SAME 80 slots (23.7 MB, cache resident: an upper bound, the round-4 figure)."""
    from vqacl_amd._lib import lib, ptr, stream_ptr
    nbuf = 8
    of = [torch.empty(B, V, cfg.feat_dim, dtype=torch.bfloat16, device=dev) for _ in range(nbuf)]
    ob = [torch.empty(B, V, 4, device=dev) for _ in range(nbuf)]
    fn = lib().vlt5_feat_gather

    def gargs(slots, i=0):
        return (ptr(store.feats), ptr(store.boxes), ptr(slots), store.capacity, ptr(of[i % nbuf]), ptr(ob[i % nbuf]), B, V, cfg.feat_dim, stream_ptr())
    a = gargs(slots_list[0])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        fn(*a)
    e0.record()
    for _ in range(200):
        fn(*a)
    e1.record()
    e1.synchronize()
    us_replay = e0.elapsed_time(e1) / 200 * 1e3
    # fresh random batches: 200 launches over rotating slot sets drawn from the whole store
    g = torch.Generator(device=dev).manual_seed(5)
    many = [torch.randint(0, store.capacity, (B,), device=dev, generator=g) for _ in range(200)]
    train = [gargs(s, i) for i, s in enumerate(many)]
    for t in train[:20]:
        fn(*t)
    torch.cuda.synchronize()
    e0.record()
    for t in train:
        fn(*t)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / len(train) * 1e3
    gbytes = 2 * B * V * (cfg.feat_dim * 2 + 16) / 1e9
    rec = {"kernel": "feat_gather_kernel", "bound": "hbm", "bytes_per_launch": int(gbytes * 1e9), "us": round(us, 2),
           "achieved": round(gbytes / (us * 1e-6), 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbytes / (us * 1e-6) / 8000.0, 4),
           "how": "200 back-to-back launches, each a different random batch of 80 out of the whole store (0.6 GB > the 256 MB Infinity Cache), "
                  "rotating output buffers, one event pair around the train",
           "us_replay_cache_resident": round(us_replay, 2), "frac_replay_cache_resident": round(gbytes / (us_replay * 1e-6) / 8000.0, 4),
           "store_images": store.capacity, "store_gb": round(store.capacity * V * (cfg.feat_dim * 2 + 16) / 1e9, 3)}
    if trace_us:
        rec["us_in_step_trace"] = trace_us[0]
        rec["frac_in_step_trace"] = round(gbytes / (trace_us[0] * 1e-6) / 8000.0, 4)
        rec["trace_source"] = trace_us[1]
    return rec


def trace_kernel_us(name):
    """Average duration (us) of kernel `name` in the committed same-build kernel trace of the bench command, or None."""
    from vqacl_amd.build import source_hash
    f = committed_profile("kernel_stats_bench_b80.txt")
    if not f:
        return None
    sha, col = None, None
    for line in open(f):
        if line.startswith("# source_sha16"):
            sha = line.split()[2]
        if line.startswith("kernel "):
            hdr = line.split()
            col = hdr.index("avg_us") - len(hdr) if "avg_us" in hdr else None
            continue
        if line.startswith("#") or col is None:
            continue
        if name in line and sha == source_hash():
            try:
                return float(line.split()[col]), f"profiles/{os.path.basename(f)}"
            except ValueError:
                return None
    return None


def main():
    args = parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, (world, args.gpus)            # (launch_plan() refuses a WORLD_SIZE that contradicts --gpus)
    rehearsal = bool(args.rehearsal)
    distributed = world > 1 or rehearsal or args.force_dist or os.environ.get("VQACL_FORCE_DIST") == "1"   # --force-dist: the RCCL path on 1 GPU
    if rehearsal:
        local = 0                                          # every rank of the rehearsal shares cuda:0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if rehearsal:
            # gloo moves CUDA tensors through the host: slow, and that is fine -- nothing of this run is a throughput figure
            import datetime
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=1800))
        else:
            # high-priority RCCL stream: the bucket collectives must start when their gradients are ready, not queue behind the
            # backward GEMMs of the compute stream (measured with tools/dp_overlap_probe.py: at normal priority the casts of a
            # bucket released mid-backward only ran after backward had finished)
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, pg_options=opts)

    from vqacl_amd import VLT5VQA, VLT5Config, FusedAdamW, reference_param_groups
    from vqacl_amd.feed import FeatureStore
    cfg = VLT5Config(dropout_rate=0.1)
    torch.manual_seed(66666)
    model = VLT5VQA(cfg, device=dev)
    model.train()
    handle = model
    dp_info = None
    if distributed:
        from vqacl_amd.parallel import DataParallelVLT5
        handle = DataParallelVLT5(model, algo=args.dp_algo)
    # --overlap-optimizer: the update of step n runs on a second stream under the forward of step n+1 (still inside the
    # timed region: the final synchronisation waits for every stream).  Measured 2 % SLOWER on MI355X (the 6.7 GB update
    # stream evicts the forward's operands from L2/MALL), so it is off by default.
    opt = FusedAdamW(reference_param_groups(model, 0.01), handle, lr=1e-4, eps=1e-6, max_grad_norm=5.0,
                     overlap=args.overlap_optimizer)
    if distributed:
        dp_info = handle.describe()
    B, L, V, T = args.batch, 20, 36, 5

    # ---- the feed: region features resident in HBM (bf16 store) before anything is timed -----------------------------------
    n_img = args.store_images
    store = FeatureStore(n_img, n_boxes=V, feat_dim=cfg.feat_dim, device=dev)
    gen = torch.Generator(device=dev).manual_seed(66666 + rank)
    for a in range(0, n_img, 256):
        store.put(list(range(a, a + 256)), torch.relu(torch.randn(256, V, cfg.feat_dim, device=dev, generator=gen)) * 1.5,
                  torch.rand(256, V, 4, device=dev, generator=gen).sort(-1).values)
    n_total = args.warmup + args.steps
    hg = torch.Generator().manual_seed(1234 + rank)
    # one host-side batch per step, as the loader would hand it over: pinned small tensors + the image ids of a fresh draw
    feeds = []
    for i in range(n_total + 16):
        small = synthetic_batch(B, L, V, T, seed=66666 + 1000 * rank + i, with_feats=False)
        small = {k: v.pin_memory() for k, v in small.items()}
        small["img_ids"] = torch.randint(0, n_img, (B,), generator=hg).tolist()
        feeds.append(small)

    def step_store(i):
        fed = {k: v for k, v in feeds[i].items() if k != "img_ids"}
        fed["feat_ref"] = store.ref(feeds[i]["img_ids"])
        res = handle.train_step(fed, 0, 0.5, 0.3)
        res["loss"].backward()
        opt.step()
        for p in model.parameters():
            p.grad = None
        return res["loss"]

    for i in range(args.warmup):
        loss = step_store(i)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, n_total):
        loss = step_store(i)
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
                                  "5 answer tokens, dropout 0.1; batch drawn fresh every step from an HBM-resident bf16 feature store",
                      "batch_per_gpu": B, "global_batch": B * world, "store_images": n_img, "parallelism": f"dp{world}",
                      "grad_exchange": (dp_info if distributed else "none")},
           "samples_per_sec_per_gpu": round(value / world, 2), "final_loss": round(final_loss, 4),
           "step_tflops_per_gpu": round(FWD_BWD_GFLOP_PER_SAMPLE * B / ms, 2),
           "step_frac_of_mfma_peak": round(FWD_BWD_GFLOP_PER_SAMPLE * B / ms / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)}
    if distributed:
        out["rccl_ranks_seen" if not rehearsal else "ranks_seen"] = dist.get_world_size()
        out["grad_exchange"] = dp_info
        out["launcher"] = "bench.py" if os.environ.get("LOCAL_WORLD_SIZE") and "TORCHELASTIC_RUN_ID" not in os.environ else "external"
        # after the timed region: every rank must hold the same weights (what the exchange is for).  Two checksums of the f32
        # master per rank, all-gathered; a rank that missed a collective or applied a different update shows up here
        flat = model.flat_params()
        mine = torch.stack([flat.double().sum(), flat.double().abs().sum()])
        seen = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(seen, mine)
        out["weights_in_sync"] = bool(all(torch.equal(x, seen[0]) for x in seen))
    if rehearsal:
        # a dress rehearsal is not a measurement: N processes time-share one GPU and the gradients travel through the host
        out["rehearsal"] = True
        out["rehearsal_samples_per_sec"] = out["value"]
        out["rehearsal_ms_per_step"] = out["ms_per_step"]
        out["value"] = None
        out["ms_per_step"] = None
        out["samples_per_sec_per_gpu"] = None
        out["step_tflops_per_gpu"] = out["step_frac_of_mfma_peak"] = None
        out["backend"] = "gloo"
        out["config"]["parallelism"] = f"dp{world} REHEARSAL: {world} rank processes share cuda:0, collectives through gloo"
        out["rehearsal_note"] = ("launcher -> N rank processes -> rendezvous -> DataParallelVLT5 -> store-fed train steps -> in-situ GEMM "
                                 "timing on every rank -> JSON relay, with the real kernels; NOT a throughput figure (value is null)")
    solo = rank == 0 and world == 1 and not distributed

    # (the side values run BEFORE the event-timed roofline pass: once a dispatch has carried timing events the runtime keeps the queue in its
    # profiling mode, launches get slower on the host side and the launch-bound decode loop measured 0.62 instead of 0.57 ms per token-step)
    if solo and not args.no_side_values:
        resident = {k: v.to(dev) for k, v in synthetic_batch(B, L, V, T, seed=66666).items()}

        def step_batch(b):
            res = handle.train_step(b, 0, 0.5, 0.3)
            res["loss"].backward()
            opt.step()
            for p in model.parameters():
                p.grad = None
        for tag, b, n in (("samples_per_sec_resident_batch", resident, 10),
                          ("samples_per_sec_pcie_inclusive", {k: v.cpu().pin_memory() for k, v in resident.items()}, 5)):
            for _ in range(2):
                step_batch(b)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                step_batch(b)
            torch.cuda.synchronize()
            out[tag] = round(n * B / (time.perf_counter() - t1), 2)
        # the gather kernel against the HBM roofline: algorithmic bytes = rows read + rows written (bf16 features + f32 boxes)
        out["feed"] = feed_figures(store, [store.slots(f["img_ids"]) for f in feeds[:21]], cfg, B, V, dev, trace_kernel_us("feat_gather_kernel"))
        # greedy decoding of the same batch shape through test_step's path (key/value cache, decode kernels): the
        # encoder + prototype retrieval + cross-K/V once, then one token per step; eos = -1 so that every row decodes the full length
        def decode_ms(bt, max_length, reps=3):
            model.eval()
            try:
                for _ in range(2):
                    model.greedy_generate(bt["input_ids"], (bt["vis_feats"], bt["boxes"]), max_length=max_length, eos_token_id=-1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    model.greedy_generate(bt["input_ids"], (bt["vis_feats"], bt["boxes"]), max_length=max_length, eos_token_id=-1)
                torch.cuda.synchronize()
                return (time.perf_counter() - t1) / reps * 1e3
            finally:
                model.train()
        wbytes = decode_weight_bytes(cfg)
        t2, t20 = decode_ms(resident, 2), decode_ms(resident, 20)
        out["decode"] = {"what": "VLT5VQA.greedy_generate (test_step), B = %d, 19 tokens per row, random weights" % B,
                         "ms_per_batch": round(t20, 3), "ms_encoder_and_first_token": round(t2, 3),
                         "ms_per_token_step": round((t20 - t2) / 18, 4), "tokens_per_sec": round(B * 19 / (t20 * 1e-3), 1),
                         "weight_bytes_per_token_step": wbytes,
                         "weight_gb_per_s": round(wbytes / ((t20 - t2) / 18 * 1e-3) / 1e9, 1),
                         "frac_of_hbm_peak": round(wbytes / ((t20 - t2) / 18 * 1e-3) / 8e12, 4)}
        # rows per launch is the lever of a launch-bound loop: the same token-step over larger evaluation batches (the reference's
        # --valid_batch_size is a free parameter of its scripts); weights are read once per step whatever the rows
        sweep = {}
        for rows in (100, 160, 320, 512):                 # (100: the reference's --valid_batch_size)
            bt = {k: v.to(dev) for k, v in synthetic_batch(rows, L, V, T, seed=4242).items()}
            a2, a20 = decode_ms(bt, 2, reps=2), decode_ms(bt, 20, reps=2)
            per = (a20 - a2) / 18
            sweep[str(rows)] = {"ms_per_token_step": round(per, 4), "tokens_per_sec": round(rows * 19 / (a20 * 1e-3), 1),
                                "weight_gb_per_s": round(wbytes / (per * 1e-3) / 1e9, 1), "ms_per_batch": round(a20, 3)}
            del bt
        out["decode"]["rows_sweep"] = sweep
    if not args.no_roofline:
        # in-situ roofline of the dominant kernel family: real steps, every GEMM dispatch timed.  Under data parallelism EVERY rank
        # runs the same extra steps (their collectives have to pair up); rank 0's records are the ones reported
        roof = insitu_gemm_roofline(lambda i: step_store(n_total + i), 2 if rehearsal else 8)
        if rank == 0:
            out["roofline"] = roof
    if solo and "roofline" in out:
        tr = trace_roofline(out["roofline"]["gflop_per_launch"] * out["roofline"]["launches_per_step"])
        if tr:
            out["roofline"].update(tr)
        pmc = committed_profile("pmc_hbm_traffic.json")
        if pmc:           # HBM bytes per GEMM launch from the committed PMC passes (tagged with the build they were taken with; not measured in this run)
            g = [r for r in json.load(open(pmc)) if any(k in r.get("kernel", "") for k in ("gemm_kernel", "skinny_kernel", "qkv_attn_fwd", "dec_attn_fwd_kernel"))]
            meta = [r for r in json.load(open(pmc)) if "source_sha16" in r]
            from vqacl_amd.build import source_hash
            if g:
                out["roofline"]["traffic"] = round(sum(r["calls"] * r["hbm_mb"] for r in g) / sum(r["calls"] for r in g) * 1e6)
                out["roofline"]["traffic_source"] = f"profiles/{os.path.basename(pmc)} (rocprofv3 --pmc, separate passes, FETCH_SIZE x2)"
                out["roofline"]["traffic_build_matches"] = bool(meta and meta[0]["source_sha16"] == source_hash())

    if solo and not args.no_side_values:
        out["configs"] = other_config_lines(dev)
    if solo and not args.no_side_values and "roofline" in out:
        warm, warm_ms = time_gemms_warm(cfg, B, L, V, T, dev)
        out["roofline"]["frac_warm"] = round(warm / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)
        out["roofline"]["gemm_ms_per_step_warm_replay"] = round(warm_ms, 3)
    if solo and not args.no_parity:
        out["parity"] = parity_vs_oracle(model, dev, B)
    if solo and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if solo and args.eager_baseline:
        del model, opt, store
        torch.cuda.empty_cache()
        out["eager_gpu_baseline"] = eager_gpu_baseline(dev, B)
    if distributed:
        dist.destroy_process_group()
    # the JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio (buffered when stdout is a pipe, it
    # would otherwise come out at exit, after the line) -- flush the C side first
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    sys.stdout.flush()


if __name__ == "__main__":
    main()
