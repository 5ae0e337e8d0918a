#!/usr/bin/env python3
"""When do the data-parallel bucket collectives finish relative to the end of backward?  (single GPU, world size 1 over RCCL)
usage: python tools/dp_overlap_probe.py   -> per merged slice: [start element, bytes, finished x ms before (-) / after (+) backward ended]"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
from oracle.ref_cpu import Cfg, synthetic_batch  # noqa: E402
from vqacl_amd import FusedAdamW, VLT5Config, VLT5VQA, reference_param_groups  # noqa: E402
from vqacl_amd.parallel import DataParallelVLT5  # noqa: E402

dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=True))
model = VLT5VQA(VLT5Config(dropout_rate=0.1), device=dev)
model.train()
dp = DataParallelVLT5(model)
opt = FusedAdamW(reference_param_groups(model, 0.01), model, lr=1e-4, eps=1e-6, max_grad_norm=5.0)
batch = {k: v.to(dev) for k, v in synthetic_batch(Cfg(), B=80, L=20, V=36, T=5, seed=1).items()}
marks = []
orig = dp._allreduce_slice


SIM_GBPS = float(os.environ.get("DP_SIM_GBPS", "0"))      # > 0: stand-in for the wire time of an N-GPU all-reduce at this algorithm
CLOCK_HZ = 2.1e9                                          # bandwidth: a one-workgroup spin kernel on the comm stream per slice


def spy(flat, a, b, **kw):
    orig(flat, a, b, **kw)
    if SIM_GBPS > 0:
        torch.cuda._sleep(int((b - a) * 2 / (SIM_GBPS * 1e9) * CLOCK_HZ))
    e = torch.cuda.Event(enable_timing=True)
    e.record()                      # on the comm stream (current inside reduce_range)
    marks.append((a, (b - a) * 2, e))


dp._allreduce_slice = spy
for it in range(6):
    marks.clear()
    start = torch.cuda.Event(enable_timing=True)
    start.record()
    dp.train_step(batch, 0, 0.5, 0.3)["loss"].backward()
    end = torch.cuda.Event(enable_timing=True)
    end.record()                    # main stream: after backward (incl. its wait for the comm stream)
    opt.step()
    for p in model.parameters():
        p.grad = None
torch.cuda.synchronize()
print(f"forward+backward {start.elapsed_time(end):.2f} ms")
for a, nbytes, e in marks:
    print(f"slice @{a:>10d} {nbytes / 1e6:7.1f} MB (bf16)  done {e.elapsed_time(end) * -1:+.3f} ms relative to the end of backward+join")
import time
t0 = time.perf_counter()
for it in range(20):
    marks.clear()
    dp.train_step(batch, 0, 0.5, 0.3)["loss"].backward()
    opt.step()
    for p in model.parameters():
        p.grad = None
torch.cuda.synchronize()
print(f"step {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms   (DP_SIM_GBPS={SIM_GBPS})")
dist.destroy_process_group()
