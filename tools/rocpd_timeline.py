#!/usr/bin/env python3
"""One optimizer step of a rocprofv3 rocpd kernel trace as a launch-by-launch timeline: start offset, duration, gap to the
previous kernel, grid.  usage: python tools/rocpd_timeline.py results.db [--step N] > profiles/timeline.txt
(a step = from one gradient-norm reduction (gnorm_final_kernel / sqnorm_final_kernel) to the next; default: the second to last one)"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    which = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else -2
    cur = db.cursor()
    kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    dcols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    namecol = "display_name" if "display_name" in kcols else "kernel_name"
    gx = "d.grid_size_x, d.workgroup_size_x, d.grid_size_y, d.grid_size_z" if "grid_size_x" in dcols else "0, 1, 1, 1"
    rows = list(cur.execute(f"""select s.{namecol}, d.start, d.end, {gx} from rocpd_kernel_dispatch d
                                join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""))
    marks = [i for i, r in enumerate(rows) if "gnorm_final_kernel" in r[0] or "sqnorm_final_kernel" in r[0]]
    a, b = marks[which - 1] if which != 0 else 0, marks[which]
    # a step starts after the optimizer of the previous one: find the last adamw launch after mark a
    j = a
    opt = lambda n: "adamw" in n or "sqnorm" in n or "gnorm" in n
    while j + 1 < b and opt(rows[j + 1][0]):
        j += 1
    seg = rows[j + 1:b + 1]
    while b + 1 < len(rows) and opt(rows[b + 1][0]):
        b += 1
        seg.append(rows[b])
    t0 = seg[0][1]
    busy = sum(r[2] - r[1] for r in seg)
    span = seg[-1][2] - t0
    print(f"# {len(seg)} launches, busy {busy / 1e3:.1f} us, span {span / 1e3:.1f} us, gaps {(span - busy) / 1e3:.1f} us")
    print(f"{'t_us':>9s} {'dur_us':>8s} {'gap_us':>7s} {'wgs':>7s}  kernel")
    prev = t0
    for name, s, e, g, w, gy, gz in seg:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*\)$", "", short).replace("void ", "")[:70]
        wgs = (g // max(w, 1)) * max(gy, 1) * max(gz, 1) if g else 0
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f} {(s - prev) / 1e3:7.2f} {wgs:7d}  {short}")
        prev = e


if __name__ == "__main__":
    main()
