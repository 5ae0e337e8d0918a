#!/usr/bin/env python3
"""Would micro-batch pipelines on separate HIP streams hide the decoder's dependent-launch latency?  (round 6 probe, not product)

The 400-row decoder chain of a B = 80 step is ~190 dependent launches of 5-8 us that leave most of the chip idle (~2.2 ms of
the 8.5 ms step), and the forward / input-gradient chains are strictly sequential within one batch.  Two half batches are independent
chains: if pipeline A's latency-bound decoder phases ran beside pipeline B's throughput-bound encoder phases, the chip would be busier.
This probe measures the CEILING of that idea with what exists: P independent models of B = 80 / P each, one stream per model, forward +
backward only (no optimizer, no shared weights, no prototype coupling -- everything a real implementation would have to add costs extra):

    enqueued:  one Python thread enqueues pipeline after pipeline per iteration (host ~2.6 ms per step: a natural phase offset)
    threads:   one Python thread per pipeline (the engine calls release the GIL)
    graphs:    dropout off, forward + backward of each pipeline captured once and replayed on its stream (no host cost at all)

against P = 1, B = 80 the same way.  Prints ms per 80 samples.

    python tools/two_pipeline_probe.py [--iters 30]
"""
import argparse
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch  # noqa: E402
from vqacl_amd import VLT5VQA, VLT5Config  # noqa: E402


def build(P, B, dropout, dev):
    models, batches, streams = [], [], []
    for p in range(P):
        torch.manual_seed(100 + p)
        m = VLT5VQA(VLT5Config(dropout_rate=dropout), device=dev)
        m.train()
        models.append(m)
        batches.append({k: v.to(dev) for k, v in synthetic_batch(B // P, seed=1 + p).items()})
        streams.append(torch.cuda.Stream(device=dev))
    return models, batches, streams


def fwd_bwd(m, b):
    res = m.train_step(b, 0, 0.5, 0.3)
    res["loss"].backward()
    for q in m.parameters():
        q.grad = None
    return res["loss"]


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--batch", type=int, default=80)
    ap.add_argument("--pipelines", type=int, nargs="*", default=[1, 2, 4])
    args = ap.parse_args()
    dev = torch.device("cuda")
    B = args.batch
    print(f"# forward + backward of {B} samples as P independent pipelines of {B}/P samples, one stream each; ms per {B} samples")
    print("P   B/P   enqueued(dropout 0.1)   threads(dropout 0.1)   graphs(dropout 0)   one pipeline alone (graph)")
    for P in args.pipelines:
        models, batches, streams = build(P, B, 0.1, dev)

        def enq():
            for m, b, s in zip(models, batches, streams):
                with torch.cuda.stream(s):
                    fwd_bwd(m, b)
        t_enq = timed(enq, args.iters)

        # one thread per pipeline, each enqueueing its own iterations back to back
        def worker(m, b, s, n):
            with torch.cuda.stream(s):
                for _ in range(n):
                    fwd_bwd(m, b)

        def run_threads(n):
            ts = [threading.Thread(target=worker, args=(m, b, s, n)) for m, b, s in zip(models, batches, streams)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        run_threads(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_threads(args.iters)
        torch.cuda.synchronize()
        t_thr = (time.perf_counter() - t0) / args.iters * 1e3
        del models, batches
        torch.cuda.empty_cache()

        models, batches, streams = build(P, B, 0.0, dev)
        graphs = []
        for m, b, s in zip(models, batches, streams):
            with torch.cuda.stream(s):
                for _ in range(2):
                    fwd_bwd(m, b)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                fwd_bwd(m, b)
            graphs.append(g)
        torch.cuda.synchronize()

        def rep():
            for g, s in zip(graphs, streams):
                with torch.cuda.stream(s):
                    g.replay()
        t_gr = timed(rep, args.iters)

        def rep_one():
            with torch.cuda.stream(streams[0]):
                graphs[0].replay()
        t_one = timed(rep_one, args.iters)
        print(f"{P}   {B // P:3d}   {t_enq:8.3f}               {t_thr:8.3f}              {t_gr:8.3f}            {t_one:8.3f}", flush=True)
        del models, batches, graphs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
