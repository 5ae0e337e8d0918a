"""Data-parallel path on CPU: world_size 2 / 4 / 8 gloo processes exercise the wrapper's collectives (weight broadcast,
flat-gradient all-reduce over the bucket range, prototype sufficient-statistics all-reduce).  The kernels need a GPU,
so gradients / pooled features are synthetic here; the numerical end-to-end DP check runs on the GPU (test_gpu_model)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import __graft_entry__ as ge
        ge.build()
        from vqacl_amd import VLT5VQA, VLT5Config
        from vqacl_amd.parallel import DataParallelVLT5
        from oracle import ref_cpu as R
        torch.manual_seed(100 + rank)                       # different init per rank: the wrapper must broadcast rank 0's
        model = VLT5VQA(VLT5Config(d_model=64, d_kv=16, num_heads=4, d_ff=128, num_layers=2, vocab_size=400, feat_dim=64),
                        device="cpu")
        before = model.flat_params().clone()
        dp = DataParallelVLT5(model)
        assert model.proto.dist_enabled and model.proto.dist_group is dp.ctrl_group and dp.ctrl_group is not None, \
            "the statistics are reduced over the wrapper's small-collective group"
        gathered = [torch.zeros_like(before) for _ in range(world)]
        dist.all_gather(gathered, model.flat_params())
        assert all(torch.equal(gathered[0], g) for g in gathered[1:]), "weights must be identical after construction"
        if rank > 0:
            assert not torch.equal(before, model.flat_params())

        # buckets: contiguous, ordered, cover exactly the gradient-bearing region, exclude prototype_fc*
        assert dp.bucket_start[0] == 0 and all(a == b for a, b in zip(dp.bucket_start[1:], dp.bucket_end[:-1]))
        used_end = max(off + n for (off, n, b, d, used) in model._pinfo.values() if used)
        assert dp.bucket_end[-1] == (used_end + 63) // 64 * 64
        unused_start = min(off for (off, n, b, d, used) in model._pinfo.values() if not used)
        assert unused_start >= used_end
        assert len(dp.bucket_end) == model._nbuckets == 2 + 2 + 2

        # gradient all-reduce = mean over ranks on the used region, untouched beyond it
        g = model.flat_grads()
        g.copy_(torch.arange(g.numel(), dtype=torch.float32) * (rank + 1))
        dp.reduce_flat(g)
        exp = torch.arange(g.numel(), dtype=torch.float32) * ((world + 1) / 2.0)       # mean over ranks of (rank + 1)
        assert torch.allclose(g[:used_end], exp[:used_end], rtol=1e-5)
        assert torch.equal(g[used_end:], (torch.arange(g.numel(), dtype=torch.float32) * (rank + 1))[used_end:])

        # reduce-scatter + all-gather of merged slices == all-reduce (algo "rs_ag" / the gradient half of "zero1"); chunks tile a slice
        dp2 = DataParallelVLT5(model, algo="rs_ag", bucket_mb=0.05)
        dflt = DataParallelVLT5(model)
        assert dp2.algo == "rs_ag" and dflt.algo == "zero1" and dflt.gather_master, "auto picks the sharded path when world | 8"
        # the slice plan (and with it the chunk a rank owns under zero1) is a function of the model alone: a step that did not come
        # out of the overlapped backward (gradient accumulation) updates the very same chunks
        plan = dflt.slice_plan()
        assert sorted(plan) == sorted((a, b) for lo, hi in dflt.release_ranges for a, b, _, _ in dflt.slices_of(lo, hi))
        assert sum(b - a for a, b in plan) == dflt.bucket_end[-1] and len(set(plan)) == len(plan)
        assert all((dflt.chunk(a, b)[1] - dflt.chunk(a, b)[0]) % 8 == 0 for a, b in plan), "chunks stay 16-byte aligned in bf16"
        # the release plan is FROZEN at construction (round 6): a later edit of the tuning record moves what the engine reports, not the
        # wrapper's slice plan / chunk ownership -- and the next backward's check refuses to run with the moved plan
        from vqacl_amd import _lib as LL
        frozen, ranges = dflt.slice_plan(), dflt.release_ranges
        old_tuning = model.tuning
        model.tuning = LL.make_tuning(wgrad_shadow=1)
        assert tuple(model.grad_release_plan()) != dflt._frozen_plan, "the engine's plan follows the tuning record"
        assert dflt.slice_plan() == frozen and dflt.release_ranges == ranges
        with pytest.raises(LL.Vlt5Error, match="release plan changed"):
            dflt.check_release_plan(model.grad_release_plan())
        model.tuning = old_tuning
        dflt.check_release_plan(model.grad_release_plan())
        # reading parameters is local in the default mode and refuses (instead of starting a collective) with a sharded master
        dflt.params_sharded = True
        from vqacl_amd._lib import Vlt5Error
        with pytest.raises(Vlt5Error):
            model.state_dict()
        dflt.params_sharded = False
        assert "shared.weight" in model.state_dict()
        # zero1 ownership: over all ranks the owned chunks tile every slice of the plan exactly once (no element updated twice, none
        # left out), whatever the world size
        owned = sorted(dflt.chunk(a, b, r) for a, b in plan for r in range(world))
        assert sum(hi - lo for lo, hi in owned) == dflt.bucket_end[-1]
        flat_cover = torch.zeros(dflt.bucket_end[-1], dtype=torch.int32)
        for lo, hi in owned:
            flat_cover[lo:hi] += 1
        assert int(flat_cover.min()) == 1 and int(flat_cover.max()) == 1
        mine_owned = [dflt.chunk(a, b) for a, b in plan]
        assert all(lo % 8 == 0 and hi % 8 == 0 for lo, hi in mine_owned), "owned chunks start and end on 16-byte boundaries of the bf16 mirror"
        model.dp = dp
        nbk = len(dp2.bucket_end)
        sl = dp2.slices_of(0, nbk)
        assert sl[0][0] == 0 and sl[-1][1] == dp2.bucket_end[-1] and all(x[1] == y[0] for x, y in zip(sl, sl[1:]))
        assert len(sl) > 1 and all((b - a) % (64 * world) == 0 or (b - a) % world == 0 for a, b, _, _ in sl)
        for a, b, _, _ in sl:
            parts = [dp2.chunk(a, b, r) for r in range(world)]
            assert parts[0][0] == a and parts[-1][1] == b and all(x[1] == y[0] for x, y in zip(parts, parts[1:]))
        g2 = torch.arange(g.numel(), dtype=torch.float32) * (rank + 1)
        for a, b, _, _ in sl:
            dp2._reduce_slice(g2, a, b)
            ca, cb = dp2.chunk(a, b)
            assert torch.allclose(g2[ca:cb], exp[ca:cb], rtol=1e-5), "own chunk holds the mean after the reduce-scatter"
        assert dp2._slices_done == [(a, b) for a, b, _, _ in sl]
        dp2._allgather_grads(g2)
        assert torch.allclose(g2[:used_end], exp[:used_end], rtol=1e-5) and dp2._slices_done == []

        # prototype statistics: N ranks x b  ==  1 process x N*b  
        gen = torch.Generator().manual_seed(7)
        pool_all = torch.randn(8, 64, generator=gen)
        ids = torch.tensor([0, 0, 1, 2, 2, 2, 9, 1])
        onehot_all = torch.zeros(8, 10).scatter_(1, ids[:, None], 1.0)
        per = 8 // world
        lo, hi = rank * per, rank * per + per
        local_proto, local_cnt = R.calculate_current_prototype(pool_all[lo:hi].unsqueeze(1), onehot_all[lo:hi])
        (proto, cnt), = model.proto._allreduce_stats((local_proto, local_cnt))
        ref_proto, ref_cnt = R.calculate_current_prototype(pool_all.unsqueeze(1), onehot_all)
        assert torch.allclose(proto, ref_proto, atol=1e-6) and torch.equal(cnt, ref_cnt)

        # evaluation: every rank predicts its shard of the questions, every rank ends up with the union (vqacl.py:585-620)
        from vqacl_amd.loop import predict

        class Stub:
            def eval(self):
                pass

            def test_step(self, batch):
                return {"pred_ans": [f"a{q}" for q in batch["question_ids"]]}
        mine = [{"question_ids": [10 * rank + k for k in range(3)]}, {"question_ids": [10 * rank + 5]}]
        merged = predict(Stub(), mine)
        assert merged == {q: f"a{q}" for r in range(world) for q in [10 * r, 10 * r + 1, 10 * r + 2, 10 * r + 5]}
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_data_parallel_wrapper_gloo(world):
    """world 2 / 4 / 8: bucket -> slice -> chunk arithmetic, slice_plan, zero1 ownership, reduce-scatter + all-gather == all-reduce,
    prototype statistics and the evaluation gather, over gloo (the RCCL run of the same wrapper: tests/test_gpu_dp2.py, bench.py)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=420) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_pick_algo_rule():
    """'auto' shards exactly when the world size divides 8 (chunks of a 64-element-aligned slice stay 16 bytes of bf16); explicit
    rs_ag / zero1 with any other world size raise; allreduce is always allowed."""
    from vqacl_amd.parallel import pick_algo
    assert pick_algo(1) == "allreduce"
    for w in (2, 4, 8):
        assert pick_algo(w) == "zero1" and pick_algo(w, "rs_ag") == "rs_ag" and pick_algo(w, "zero1") == "zero1"
    for w in (3, 5, 6, 7, 16, 64):
        assert pick_algo(w) == "allreduce" and pick_algo(w, "allreduce") == "allreduce"
        for algo in ("rs_ag", "zero1"):
            with pytest.raises(ValueError):
                pick_algo(w, algo)
    with pytest.raises(ValueError):
        pick_algo(2, "ring")
    with pytest.raises(ValueError):
        pick_algo(0)
