#!/usr/bin/env python3
"""What the data-parallel wrapper puts on the links per GPU and step, collective by collective, from the model's own layout (no GPU needed:
the parameter layout and the gradient release plan come from the library).  A MODEL, not a measurement: bytes are exact, times assume a ring
bus bandwidth (--busbw GB/s, default 300: the xGMI figure a large RCCL all-gather reaches on an 8-GPU node is the first thing to measure).

    python tools/link_budget.py [--world 8] [--busbw 300] [--large] [--no-master]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--busbw", type=float, default=300.0, help="ring bus bandwidth in GB/s")
    ap.add_argument("--large", action="store_true")
    ap.add_argument("--no-master", action="store_true", help="gather_master=False")
    ap.add_argument("--bucket-mb", type=float, default=128)
    args = ap.parse_args()
    import torch
    from vqacl_amd import VLT5VQA, VLT5Config
    kw = dict(d_model=1024, num_heads=16, d_ff=4096, num_layers=24) if args.large else {}
    m = VLT5VQA(VLT5Config(**kw), device=torch.device("cpu"))
    ends = {}
    for name, (off, n, bucket, decay, used) in m._pinfo.items():
        if used:
            ends[bucket] = max(ends.get(bucket, 0), (off + n + 63) // 64 * 64)
    bend = [ends[b] for b in sorted(ends)]
    bstart = [0] + bend[:-1]
    N, f = args.world, (args.world - 1) / args.world
    bb = int(args.bucket_mb * (1 << 20))

    from vqacl_amd.parallel import merge_buckets

    def slices(lo, hi):
        return [(a, b) for a, b, _, _ in merge_buckets(bstart, bend, lo, hi, bb)]
    names = {0: "after vlt5_decoder_bwd", 1: "inside / after vlt5_encoder_bwd"}
    last_a = bstart[-1]
    total = dict(rs=0.0, ag=0.0, master=0.0)
    print(f"# {'VL-T5-large' if args.large else 'VL-T5-base'}, world {N}, zero1, bf16 buckets, slices merged to >= {args.bucket_mb:g} MB (f32 size), "
          f"{bend[-1] / 1e6:.1f} M gradient elements; times at {args.busbw:g} GB/s of ring bus bandwidth")
    print("# reduce-scatter during backward (bytes on the links per GPU = (N-1)/N x slice bytes):")
    for phase, lo, hi in m.grad_release_plan():
        for a, b in slices(lo, hi):
            mb = (b - a) * 2 / 1e6
            total["rs"] += mb * f
            print(f"   {names[phase]:34s} elements [{a:>10d}, {b:>10d})  {mb:7.1f} MB bf16  -> {mb * f:7.1f} MB on the links, {mb * f / args.busbw:6.3f} ms")
    print("# all-gather of the updated parameters after the sharded step, in the order the next forward reads them:")
    plan = [s for _, lo, hi in m.grad_release_plan() for s in slices(lo, hi)]
    for a, b in sorted(plan, key=lambda s: -s[0]):
        sh = (b - a) * 2 / 1e6
        extra = (b - a) * 4 / 1e6 if a >= last_a else 0.0
        total["ag"] += (sh + extra) * f
        print(f"   bf16 shadow{' + f32 master (embeddings / norms)' if extra else '':33s} [{a:>10d}, {b:>10d})  {sh + extra:7.1f} MB -> {(sh + extra) * f:7.1f} MB, {(sh + extra) * f / args.busbw:6.3f} ms")
    if not args.no_master:
        mm = last_a * 4 / 1e6
        total["master"] = mm * f
        print(f"# f32 master of the layer buckets behind them (gather_master=True; only the next optimizer step waits for it): {mm:7.1f} MB -> {mm * f:7.1f} MB, {mm * f / args.busbw:6.3f} ms")
    t = sum(total.values())
    print(f"# per GPU and step: reduce-scatter {total['rs'] / 1e3:.2f} GB + parameter all-gather {total['ag'] / 1e3:.2f} GB + master {total['master'] / 1e3:.2f} GB "
          f"= {t / 1e3:.2f} GB = {t / args.busbw:.2f} ms of link time at {args.busbw:g} GB/s")


if __name__ == "__main__":
    main()
