#!/usr/bin/env python3
"""What would replaying the train step from a HIP graph buy at small batches?  (VERDICT r04 item 5 / SURVEY 7 step 7)

Probe, not product: with dropout OFF (the dropout seeds are launch arguments by value; a replayed graph would repeat one mask) the whole
forward + backward of VLT5VQA.train_step is captured once per batch size with torch.cuda.graph on static inputs and replayed; the
optimizer (its step count is a launch argument too) stays enqueued.  Prints, per batch size: the enqueued step (host-bound or not),
the replayed step, the host time of the enqueue, and the sum of kernel durations a profiler would see (the floor of either).

    python tools/step_graph_probe.py [B ...]        # default 4 8 16 32 80
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch  # noqa: E402
from vqacl_amd import VLT5VQA, VLT5Config, FusedAdamW, reference_param_groups  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [4, 8, 16, 32, 80]
    dev = torch.device("cuda")
    model = VLT5VQA(VLT5Config(dropout_rate=0.0), device=dev)
    model.train()
    opt = FusedAdamW(reference_param_groups(model, 0.01), model, lr=1e-4, eps=1e-6, max_grad_norm=5.0)
    print("B   enqueued ms/step (host ms)   graph-replayed fwd+bwd + enqueued AdamW ms/step   fwd+bwd replay alone   AdamW alone")
    for B in sizes:
        batch = {k: v.to(dev) for k, v in synthetic_batch(B, seed=1).items()}

        def fwd_bwd():
            res = model.train_step(batch, 0, 0.5, 0.3)
            res["loss"].backward()
            return res["loss"]

        def clear():
            for p in model.parameters():
                p.grad = None
        for _ in range(4):
            fwd_bwd()
            opt.step()
            clear()
        torch.cuda.synchronize()
        n = 30
        host = 0.0
        t0 = time.perf_counter()
        for _ in range(n):
            h0 = time.perf_counter()
            fwd_bwd()
            opt.step()
            clear()
            host += time.perf_counter() - h0
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n * 1e3
        host = host / n * 1e3
        # capture forward + backward (the gradients land in the flat buffer the optimizer reads: same addresses every replay)
        g = torch.cuda.CUDAGraph()
        clear()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            loss = fwd_bwd()
        for _ in range(3):
            g.replay()
            opt.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
            opt.step()
        torch.cuda.synchronize()
        replay = (time.perf_counter() - t0) / n * 1e3
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        only = (time.perf_counter() - t0) / n * 1e3
        t0 = time.perf_counter()
        for _ in range(n):
            opt.step()
        torch.cuda.synchronize()
        adam = (time.perf_counter() - t0) / n * 1e3
        print(f"{B:3d}   {eager:7.3f} ({host:6.3f})               {replay:7.3f}                                   {only:7.3f}               {adam:7.3f}   loss {float(loss.detach()):.4f}", flush=True)
        clear()
        del g, batch


if __name__ == "__main__":
    main()
