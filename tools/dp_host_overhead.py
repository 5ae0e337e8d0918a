#!/usr/bin/env python3
"""Host enqueue time against wall clock per train step, plain and through DataParallelVLT5 over RCCL with one rank: is the +0.7-0.85 ms of
wall clock of `bench.py --force-dist` the host falling behind the device?   python tools/dp_host_overhead.py"""
import os
import socket
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch  # noqa: E402
from vqacl_amd import VLT5VQA, VLT5Config, FusedAdamW, reference_param_groups  # noqa: E402


def measure(handle, model, opt, batch, n=30):
    def step():
        res = handle.train_step(batch, 0, 0.5, 0.3)
        res["loss"].backward()
        opt.step()
        for p in model.parameters():
            p.grad = None
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    host = []
    free_run = {"forward": 0.0, "backward": 0.0, "optimizer": 0.0}          # host time of each call in the free-running loop (a call that blocks shows here)
    a0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    t_all = time.perf_counter()
    for _ in range(n):
        t0 = time.perf_counter()
        res = handle.train_step(batch, 0, 0.5, 0.3)
        t1 = time.perf_counter()
        res["loss"].backward()
        t2 = time.perf_counter()
        opt.step()
        t3 = time.perf_counter()
        for p in model.parameters():
            p.grad = None
        host.append(time.perf_counter() - t0)
        free_run["forward"] += (t1 - t0) / n * 1e3
        free_run["backward"] += (t2 - t1) / n * 1e3
        free_run["optimizer"] += (t3 - t2) / n * 1e3
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t_all) / n
    print("      free-running host time per call: " + ", ".join(f"{k} {v:.2f} ms" for k, v in free_run.items()) +
          f"; device allocations during the {n} steps: {torch.cuda.memory_stats().get('num_device_alloc', 0) - a0}")
    host.sort()
    # phases of the host time of one step (synchronised in between, so only the ENQUEUE cost of each phase is seen)
    ph = {}
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); res = handle.train_step(batch, 0, 0.5, 0.3); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter(); res["loss"].backward(); t3 = time.perf_counter()
        torch.cuda.synchronize(); t4 = time.perf_counter(); opt.step(); t5 = time.perf_counter()
        for p in model.parameters():
            p.grad = None
        for k, v in (("forward", t1 - t0), ("backward", t3 - t2), ("optimizer", t5 - t4)):
            ph.setdefault(k, []).append(v)
    return host[len(host) // 2] * 1e3, wall * 1e3, {k: sorted(v)[len(v) // 2] * 1e3 for k, v in ph.items()}


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    batch = {k: v.to(dev) for k, v in synthetic_batch(80, seed=1).items()}
    order = [a for a in sys.argv[1:] if a != "--no-plain"] or ["allreduce", "zero1"]
    if "--no-plain" not in sys.argv:
        model = VLT5VQA(VLT5Config(dropout_rate=0.1), device=dev)
        model.train()
        opt = FusedAdamW(reference_param_groups(model, 0.01), model, lr=1e-4, eps=1e-6, max_grad_norm=5.0)
        h, w, ph = measure(model, model, opt, batch)
        print(f"plain:        host enqueue {h:6.2f} ms per step, wall {w:6.2f} ms   (enqueue by phase: " + ", ".join(f"{k} {v:.2f}" for k, v in ph.items()) + ")")
        del model, opt
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=opts)
    from vqacl_amd.parallel import DataParallelVLT5
    if os.environ.get("PROBE_SLEEP_AFTER_INIT"):
        time.sleep(float(os.environ["PROBE_SLEEP_AFTER_INIT"]))
    keep = []
    if os.environ.get("PROBE_DUMMY_STREAMS"):
        keep = [torch.cuda.Stream(priority=-1) for _ in range(int(os.environ["PROBE_DUMMY_STREAMS"]))]      # shift torch's high-priority stream pool
    for algo in order:
        model = VLT5VQA(VLT5Config(dropout_rate=0.1), device=dev)
        model.train()
        dp = DataParallelVLT5(model, algo=algo)
        opt = FusedAdamW(reference_param_groups(model, 0.01), dp, lr=1e-4, eps=1e-6, max_grad_norm=5.0)
        h, w, ph = measure(dp, model, opt, batch)
        print(f"dp {algo:9s}: host enqueue {h:6.2f} ms per step, wall {w:6.2f} ms   (enqueue by phase: " + ", ".join(f"{k} {v:.2f}" for k, v in ph.items()) + ")")
        del model, opt, dp
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
