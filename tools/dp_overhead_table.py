#!/usr/bin/env python3
"""What the data-parallel machinery costs at world size 1, item by item: per-kernel time per optimizer step of the plain step against the
same step through DataParallelVLT5 over RCCL with one rank (two rocprofv3 kernel-trace summaries of tools/rocpd_stats.py), each item
classed as a single-rank stand-in that a real collective replaces at N > 1, or as work that stays.
usage: python tools/dp_overhead_table.py plain_stats.txt dist_stats.txt [plain_ms dist_ms]"""
import re
import sys

CLASS = [
    ("__amd_rocclr_copyBuffer", "vanishes", "RCCL's one-rank stand-in for a collective (device-to-device copy of the slice); at N > 1 the reduce-scatter / all-reduce kernel on the comm stream"),
    ("__amd_rocclr_fillBufferAligned", "vanishes", "RCCL one-rank bookkeeping fills"),
    ("adamw_kernel<true>", "stays (cheaper)", "AdamW reading the reduced bf16 bucket x 1/world from the staging buffer: 26 instead of 30 B/param; 1/N of it under zero1"),
    ("adamw_kernel<false>", "stays (cheaper)", "(the plain step's AdamW, replaced by the row above)"),
    ("sqnorm_kernel<true>", "stays, 1/N under zero1", "gradient norm of the REDUCED gradients: a pass over the bf16 staging buffer (the per-tile shares of the weight-gradient GEMMs are local sums and cannot be used)"),
    ("sqnorm_kernel<false>", "stays, 1/N under zero1", "(the plain step's norm over the ranges no GEMM writes)"),
    ("gemm_kernel<128, 128, 2, 2, true, true, 2>", "stays (by design)", "the decoder's six batched weight gradients as launches of their own: they release the decoder's gradient buckets at the END of the decoder phase, 1.5 ms before the encoder's, so that 198 MB of the exchange start early (riding in the encoder's launches instead: tools/experiments/dp_shadow_wgrads.patch, -0.09 ms kernel time, no wall-clock gain at world 1)"),
    ("gemm_kernel<256, 256, 2, 4, true, true, 2>", "stays", "weight-gradient epilogues also write the bf16 staging copy of the bucket (saves the cast pass of every layer bucket)"),
    ("cast_kernel", "stays", "bf16 cast of the last bucket (embeddings, norms, visual embedding: scatter-added, no GEMM epilogue to mirror them)"),
    ("retrieve_kernel", "stays", "prototype head as separate launches around the statistics all-reduce"),
    ("class_mean_kernel", "stays", "prototype head as separate launches around the statistics all-reduce"),
    ("proto_normalize_kernel", "stays", "prototype head as separate launches around the statistics all-reduce"),
    ("proto_update_kernel", "stays", "prototype head as separate launches around the statistics all-reduce"),
    ("retrieve2_kernel", "stays", "(the plain step's fused prototype head)"),
    ("proto_row_kernel", "stays", "(the plain step's fused prototype head)"),
    ("at::native", "stays", "torch-side small ops of the wrapper (scalar all-reduce operands, statistics concatenation)"),
]


def load(path):
    steps, d = None, {}
    for line in open(path):
        m = re.match(r"# (\d+) optimizer steps", line)
        if m:
            steps = int(m.group(1))
        m = re.match(r"(.{90})\s+(\d+)\s+([\d.]+)\s+([\d.]+)", line)
        if m and not line.startswith("kernel "):
            d[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return steps, d


def main():
    sa, a = load(sys.argv[1])
    sb, b = load(sys.argv[2])
    rows, tot = [], {"vanishes": 0.0, "stays": 0.0}
    once = []
    for k in sorted(set(a) | set(b)):
        ca_, cb_ = a.get(k, (0, 0.0))[0], b.get(k, (0, 0.0))[0]
        if "at::native" in k and ((ca_ % sa) or (cb_ % sb)):
            # torch-side kernels whose launch count is not a multiple of the optimizer steps run once per PROCESS (feature-store fill,
            # model initialisation, the bench's weights-in-sync check after the timed region): no part of a step, and the two traces
            # cover different step counts -- listed, not summed
            once.append((k, ca_, a.get(k, (0, 0.0))[1], cb_, b.get(k, (0, 0.0))[1]))
            continue
        ta, tb = a.get(k, (0, 0.0))[1] / sa, b.get(k, (0, 0.0))[1] / sb
        if abs(tb - ta) < 0.004:
            continue
        cls, why = next(((c, w) for pat, c, w in CLASS if pat in k), ("stays", "(same launches, time differs: shared HBM / L2 with the communication stream)"))
        rows.append((tb - ta, k, a.get(k, (0, 0))[0] / sa, ta, b.get(k, (0, 0))[0] / sb, tb, cls, why))
        tot["vanishes" if cls == "vanishes" else "stays"] += tb - ta
    rows.sort(key=lambda r: -abs(r[0]))
    print(f"# kernel time per optimizer step, ms: plain step ({sys.argv[1].split('/')[-1]}, {sa} steps) against the step through DataParallelVLT5 over RCCL, world size 1")
    print(f"# ({sys.argv[2].split('/')[-1]}, {sb} steps); rows below 4 us of difference omitted")
    print(f"{'kernel':58s} {'plain n':>7s} {'ms':>7s} {'dist n':>7s} {'ms':>7s} {'delta':>7s}  class / why")
    for dlt, k, ca, ta, cb, tb, cls, why in rows:
        print(f"{k[:58]:58s} {ca:7.1f} {ta:7.3f} {cb:7.1f} {tb:7.3f} {dlt:+7.3f}  {cls}: {why}")
    skip = {k for k, *_ in once}
    ka, kb = sum(v[1] for k, v in a.items() if k not in skip) / sa, sum(v[1] for k, v in b.items() if k not in skip) / sb
    print(f"# kernel time per step (per-process torch kernels excluded): {ka:.3f} -> {kb:.3f} ms ({kb - ka:+.3f}); of the listed rows {tot['vanishes']:+.3f} ms are "
          f"single-rank stand-ins, {tot['stays']:+.3f} ms stay")
    print("# once per process, not per step (calls / total ms in the plain trace | in the dist trace):")
    for k, ca_, ta_, cb_, tb_ in sorted(once, key=lambda r: -(r[2] + r[4])):
        print(f"#   {k[:70]:70s} {ca_:5d} {ta_:8.3f} | {cb_:5d} {tb_:8.3f}")
    if len(sys.argv) > 4:
        pa, pb = float(sys.argv[3]), float(sys.argv[4])
        print(f"# wall clock per step (same box, un-profiled): {pa:.3f} -> {pb:.3f} ms ({pb - pa:+.3f}); the part beyond the kernel-time difference is the second hardware "
              f"queue (every kernel that runs on the communication stream beside the chain delays the chain's dependent launches: tools/event_cost.py)")


if __name__ == "__main__":
    main()
