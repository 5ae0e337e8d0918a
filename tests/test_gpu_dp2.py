"""Two (and four, and eight) data-parallel RANKS on ONE GPU (both processes use cuda:0, collectives through gloo on CUDA tensors): the whole
multi-rank path of `vqacl_amd.parallel` -- weight broadcast, gradient buckets released by the engine's HIP events, the comm
stream, the prototype-statistics all-reduce, the fused optimizer after the reduction -- runs with the real kernels and is
compared against ONE process on the concatenated batch ("N ranks x b == 1 process x N*b").  RCCL itself cannot
place two ranks on one device; the single-rank RCCL run of the same code path is in test_gpu_model.py, and the collective
semantics on CPU in test_dp_cpu.py.
"""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(ocfg, params, dev):
    from vqacl_amd import VLT5VQA, VLT5Config
    cfg = VLT5Config(d_model=ocfg.d_model, d_kv=ocfg.d_kv, num_heads=ocfg.num_heads, d_ff=ocfg.d_ff, num_layers=ocfg.num_layers,
                     num_decoder_layers=ocfg.num_decoder_layers, vocab_size=ocfg.vocab_size, feat_dim=ocfg.feat_dim,
                     dropout_rate=0.0, n_ques=ocfg.n_ques, n_cate=ocfg.n_cate)
    m = VLT5VQA(cfg, device=dev)
    m.load_state_dict({k: v.detach() for k, v in params.items()}, strict=False)
    return m


def _worker(rank, world, port, q, grad_dtype, algo="allreduce", backend="gloo"):
    lazy = algo == "zero1-lazy"             # zero1 with gather_master=False: the f32 master stays sharded until consolidate()
    algo = "zero1" if lazy else algo
    try:
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HSA_ENABLE_IPC_MODE_LEGACY="0")
        dev = torch.device("cuda", rank if backend == "nccl" else 0)      # RCCL: one GPU per rank; gloo: both ranks on cuda:0
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import ref_cpu as R
        from vqacl_amd import FusedAdamW, reference_param_groups
        from vqacl_amd.parallel import DataParallelVLT5
        ocfg = R.tiny_cfg()
        params = R.init_params(ocfg, seed=77)
        full = R.synthetic_batch(ocfg, B=max(8, 2 * world), L=12, V=36, T=4, seed=5)     # >= 2 samples per rank (the accumulation step halves them)
        b = full["input_ids"].shape[0] // world
        mine = {k: (v[rank * b:(rank + 1) * b] if torch.is_tensor(v) else v) for k, v in full.items()}

        # every rank starts from DIFFERENT weights: the wrapper must install rank 0's
        wrong = {k: (v + 0.01 * rank) for k, v in params.items()}
        model = _model(ocfg, wrong, dev)
        model.train()
        dp = DataParallelVLT5(model, bucket_mb=0.05,          # tiny buckets: several collectives interleaved with backward
                              grad_dtype=getattr(torch, grad_dtype), algo=algo, gather_master=not lazy)
        assert dp.algo == algo
        opt = FusedAdamW(reference_param_groups(model, 0.01), model, lr=1e-3, eps=1e-6, max_grad_norm=5.0)
        losses, grads0 = [], None
        assert dp.sharded_optimizer == (algo == "zero1")
        for it in range(3):
            res = dp.train_step(mine, 0, 0.5, 0.3)
            res["loss"].backward()
            assert dp.shards_valid == (algo == "zero1")
            if it == 0:
                torch.cuda.synchronize()
                if algo == "zero1" and world > 1:       # a silent collective behind a gradient read would hang a rank-0-only caller: it raises
                    from vqacl_amd._lib import Vlt5Error
                    try:
                        model.flat_grads()
                        raise AssertionError("flat_grads() under zero1 must refuse to start a hidden collective")
                    except Vlt5Error:
                        pass
                grads0 = model.flat_grads(collective=True).clone()       # (zero1: all-gathers the reduced chunks -- collective; the step below is still the sharded one)
                assert dp.shards_valid == (algo == "zero1")
            opt.step()
            assert dp.params_sharded == (lazy and world > 1)
            for p in model.parameters():
                p.grad = None
            losses.append(float(res["loss"].detach()))
        # a fourth step with gradient accumulation over two half batches (the non-overlapped reduction path): under zero1 it must
        # still be the sharded update -- the Adam moments of the chunks a rank does not own are stale since its first sharded step
        for half in (slice(0, b // 2), slice(b // 2, b)):
            part = {k: (v[half] if torch.is_tensor(v) else v) for k, v in mine.items()}
            dp.train_step(part, 0, 0.5, 0.3)["loss"].backward()
        opt.step()
        for p in model.parameters():
            p.grad = None
        torch.cuda.synchronize()
        if lazy:
            # the master is sharded: reading parameters must fail loudly, not start a collective behind the caller's back
            from vqacl_amd._lib import Vlt5Error
            for fn in (model.state_dict, model.flat_params):
                try:
                    fn()
                    raise AssertionError("a sharded master must not be readable before consolidate()")
                except Vlt5Error:
                    pass
            dp.consolidate()
        elif rank == 0:
            # the reference checkpoints on rank 0 ONLY (vqacl.py:413-414) while the other ranks walk on: state_dict() must be a
            # local call in the default mode (a collective here would hang this test into its timeout)
            sd = model.state_dict()
            assert torch.isfinite(sd["decoder.block.0.layer.0.SelfAttention.q.weight"]).all()
        flat = model.flat_params().clone()
        protoQ = model.Q_prototype.clone()

        # all ranks hold the same weights and prototypes after three steps
        other = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(other, flat)
        assert all(torch.equal(o, other[0]) for o in other), "ranks diverged"
        pq = [torch.zeros_like(protoQ) for _ in range(world)]
        dist.all_gather(pq, protoQ)
        assert all(torch.allclose(o, pq[0], atol=1e-6) for o in pq), "prototypes diverged"

        if rank == 0:
            # one process on the concatenated batch
            ref = _model(ocfg, params, dev)
            ref.train()
            ropt = FusedAdamW(reference_param_groups(ref, 0.01), ref, lr=1e-3, eps=1e-6, max_grad_norm=5.0)
            rl, rg0 = [], None
            for it in range(3):
                r = ref.train_step(full, 0, 0.5, 0.3)
                r["loss"].backward()
                if it == 0:
                    torch.cuda.synchronize()
                    rg0 = ref.flat_grads().clone()
                ropt.step()
                for p in ref.parameters():
                    p.grad = None
                rl.append(float(r["loss"].detach()))
            for lo in (0, b // 2):        # the accumulation step: micro-batch = the same halves of every rank's samples
                sel = torch.cat([torch.arange(r_ * b + lo, r_ * b + lo + b // 2) for r_ in range(world)])
                part = {k: (v[sel] if torch.is_tensor(v) else v) for k, v in full.items()}
                ref.train_step(part, 0, 0.5, 0.3)["loss"].backward()
            ropt.step()
            for p in ref.parameters():
                p.grad = None
            torch.cuda.synchronize()

            def cosine(a, c):
                return float(torch.dot(a.flatten(), c.flatten()) / (a.norm() * c.norm()).clamp(min=1e-30))
            assert cosine(grads0, rg0) > 0.9995, cosine(grads0, rg0)                  # mean of rank gradients == full-batch gradient
            assert abs(float(grads0.norm() / rg0.norm()) - 1.0) < 1e-2
            assert cosine(flat, ref.flat_params()) > 0.999999
            assert float((flat - ref.flat_params()).abs().max()) < 7e-3               # lr 1e-3 x 4 steps bounds any difference
            assert torch.allclose(protoQ, ref.Q_prototype, atol=2e-2), float((protoQ - ref.Q_prototype).abs().max())
            # the rank-local loss is the mean over b, the reference's over N*b: rank 0's differs, the mean over ranks matches
        lt = torch.tensor(losses, device=dev)
        dist.all_reduce(lt)
        if rank == 0:
            for a, c in zip((lt / world).tolist(), rl):
                assert abs(a - c) < 2e-2, (lt.tolist(), rl)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def _long_worker(rank, world, port, q, steps):
    """`steps` optimizer steps through the sharded default path with what a continual run does in between: a task switch (the prototype
    state machine's other branches), a NEW optimizer half-way (the Trainer builds one per category group: fresh Adam moments against
    weights whose non-owned master chunks arrived by all-gather), a rehearsal-style batch with mixed question types -- against one
    process on the concatenated batches."""
    try:
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import ref_cpu as R
        from vqacl_amd import FusedAdamW, reference_param_groups
        from vqacl_amd.parallel import DataParallelVLT5
        ocfg = R.tiny_cfg()
        params = R.init_params(ocfg, seed=78)
        b = 2
        batches, tasks = [], []
        for i in range(steps):
            task = 0 if i < steps // 3 else 1
            bt = R.synthetic_batch(ocfg, B=b * world, L=12, V=36, T=4, seed=100 + i, task_id=task)
            if task == 1 and i % 2 == 1:                      # rehearsal batch: question types of the earlier task, run under the current task id
                bt["ques_labels"] = R.synthetic_batch(ocfg, B=b * world, L=12, V=36, T=4, seed=100 + i, task_id=0)["ques_labels"]
            batches.append(bt)
            tasks.append(task)

        def run(model, handle, sl):
            opt, losses = None, []
            for i, (bt, task) in enumerate(zip(batches, tasks)):
                if i in (0, steps // 2):
                    opt = FusedAdamW(reference_param_groups(model, 0.01), handle, lr=1e-3, eps=1e-6, max_grad_norm=5.0)
                mine = {k: v[sl] for k, v in bt.items()}
                res = handle.train_step(mine, task, 0.5, 0.3)
                res["loss"].backward()
                opt.step()
                for p in model.parameters():
                    p.grad = None
                losses.append(float(res["loss"].detach()))
            torch.cuda.synchronize()
            return losses
        model = _model(ocfg, params, dev)
        model.train()
        dp = DataParallelVLT5(model, bucket_mb=0.05)
        assert dp.algo == "zero1" and dp.grad_dtype is torch.bfloat16
        losses = run(model, dp, slice(rank * b, (rank + 1) * b))
        flat = model.flat_params().clone()
        other = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(other, flat)
        assert all(torch.equal(o, other[0]) for o in other), "ranks diverged over the long run"
        lt = torch.tensor(losses, device=dev)
        dist.all_reduce(lt)
        if rank == 0:
            ref = _model(ocfg, params, dev)
            ref.train()
            rl = run(ref, ref, slice(0, b * world))
            mean = (lt / world).tolist()
            worst = max(abs(a - c) for a, c in zip(mean, rl))
            assert worst < 5e-2, (worst, mean, rl)
            assert max(abs(a - c) for a, c in zip(mean[:6], rl[:6])) < 1e-2
            a, c = flat, ref.flat_params()
            cosw = float(torch.dot(a, c) / (a.norm() * c.norm()))
            assert cosw > 0.99999, cosw
            assert torch.allclose(model.Q_prototype, ref.Q_prototype, atol=5e-2) and torch.allclose(model.V_prototype, ref.V_prototype, atol=5e-2)
            assert all(x == x and x < 20.0 for x in rl)                                    # finite (fresh random batches every step: no trend to assert)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4])
def test_many_steps_of_the_sharded_path_track_one_process(world):
    """24 optimizer steps at world 2 and 4 (zero1, bf16 buckets, real kernels) with a task switch, rehearsal-style batches and a new optimizer
    half-way: the mean loss of the ranks tracks one process on the concatenated batches step by step, the ranks end bit-identical, the weights
    agree with the single process to cosine > 0.99999."""
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_long_worker, args=(r, world, port, q, 24)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=600) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _run_two(grad_dtype, algo, backend, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, grad_dtype, algo, backend)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=420) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


@pytest.mark.timeout(600)
@pytest.mark.parametrize("grad_dtype,algo", [("float32", "allreduce"), ("bfloat16", "allreduce"), ("float32", "rs_ag"),
                                             ("bfloat16", "zero1"), ("float32", "zero1"), ("bfloat16", "zero1-lazy")])
def test_two_ranks_on_one_gpu_equal_one_process_on_the_concatenated_batch(grad_dtype, algo):
    assert torch.cuda.is_available()
    _run_two(grad_dtype, algo, "gloo")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,grad_dtype,algo", [(4, "bfloat16", "zero1"), (8, "bfloat16", "zero1"), (4, "float32", "zero1"),
                                                   (8, "bfloat16", "allreduce"), (4, "bfloat16", "rs_ag"), (8, "bfloat16", "zero1-lazy")])
def test_four_and_eight_ranks_on_one_gpu_equal_one_process_on_the_concatenated_batch(world, grad_dtype, algo):
    """World sizes 4 and 8 (the sizes the scaling bench runs) with the real kernels: chunk ownership of the sharded optimizer (every
    slice cut into 4 / 8 chunks of whole 16-byte groups), parameter all-gathers ordered against the next forward, the prototype
    statistics summed over 4 / 8 ranks -- N ranks x b samples == one process x N*b samples, and every rank ends with the same weights."""
    assert torch.cuda.is_available()
    _run_two(grad_dtype, algo, "gloo", world=world)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("grad_dtype,algo", [("bfloat16", "allreduce"), ("bfloat16", "zero1")])
def test_two_gpus_rccl_equal_one_process_on_the_concatenated_batch(grad_dtype, algo):
    """The same check over RCCL with one GPU per rank -- runs wherever two GPUs are visible (the build's GPU box has one: skipped
    there).  Children are spawned before this process touches the GPU (device_count() does not initialise it)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs for a two-rank RCCL group")
    _run_two(grad_dtype, algo, "nccl")


@pytest.mark.timeout(600)
def test_wrapper_step_does_not_depend_on_when_the_process_group_was_created():
    """Round 5 found a wrapper step three times slower (26 ms against 9.2) for the whole life of the model whenever the GPU had been touched BEFORE
    dist.init_process_group(): the wrapper's HIGH-priority communication stream sitting on pending bucket-event waits.  The stream is normal
    priority now; this guards the fix with a ratio, not an absolute time: the late-init order must run within 1.3x of the init-first order."""
    import subprocess
    assert torch.cuda.is_available()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VQACL_COMM_PRIORITY")}
    res = {}
    for mode in ("init_first", "touch_then_init"):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_first_model_probe.py"), mode], capture_output=True, text=True, timeout=280,
                           env=env, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][-1].split()
        res[mode] = float(line[2])
        assert line[3] == "0", f"the wrapper's communication stream must be normal priority by default (got {line[3]})"
    assert res["touch_then_init"] < 1.3 * res["init_first"], res
