#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls / total / average / share, like `--stats`.
usage: python tools/rocpd_stats.py results.db [--steps N] > profiles/summary.txt   (steps default: counted from the trace)"""
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqacl_amd.build import source_hash  # noqa: E402


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else None
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    namecol = "display_name" if "display_name" in kcols else "kernel_name"
    q = f"""select s.{namecol}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
            from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
            group by s.{namecol} order by 3 desc"""
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows)
    span = list(cur.execute("select min(start), max(end) from rocpd_kernel_dispatch"))[0]
    print(f"# source_sha16 {source_hash()}   (vqacl_amd.build.source_hash() of the tree this trace was taken with)")
    print(f"# kernels: {sum(r[1] for r in rows)} dispatches, {total / 1e6:.3f} ms busy, span {(span[1] - span[0]) / 1e6:.3f} ms")
    if steps is None:                                   # one "final" reduction of the gradient norm per optimizer step (either path)
        steps = sum(r[1] for r in rows if "gnorm_final_kernel" in r[0] or "sqnorm_final_kernel" in r[0]) or None
    if steps:
        print(f"# {steps} optimizer steps in the trace (timed + warm-up + the PCIe-inclusive side loop): "
              f"{total / 1e6 / steps:.3f} ms kernel time per step")
    print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'%':>6s}")
    for name, n, tot, mn, mx in rows:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*\)$", "", short)[:90]
        print(f"{short:90s} {n:7d} {tot / 1e6:10.3f} {tot / n / 1e3:9.2f} {mn / 1e3:8.2f} {mx / 1e3:8.2f} {100.0 * tot / total:6.2f}")


if __name__ == "__main__":
    main()
