#!/usr/bin/env python3
"""A window of a rocprofv3 rocpd kernel trace: the dispatches from the k-th launch of a marker kernel to the next one, launch by launch
(start offset, duration, gap to the previous kernel, workgroups).  usage: python tools/rocpd_window.py results.db <marker substring> [k]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2]
    k = int(sys.argv[3]) if len(sys.argv) > 3 else -3
    cur = db.cursor()
    kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    dcols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    namecol = "display_name" if "display_name" in kcols else "kernel_name"
    gx = "d.grid_size_x, d.workgroup_size_x, d.grid_size_y, d.grid_size_z" if "grid_size_x" in dcols else "0, 1, 1, 1"
    rows = list(cur.execute(f"""select s.{namecol}, d.start, d.end, {gx} from rocpd_kernel_dispatch d
                                join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""))
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    a, b = marks[k], marks[k + 1]
    seg = rows[a:b + 1]
    t0 = seg[0][1]
    busy = sum(r[2] - r[1] for r in seg[1:])
    span = seg[-1][2] - seg[0][2]
    print(f"# {len(seg) - 1} launches between two `{marker}` launches: busy {busy / 1e3:.1f} us, span {span / 1e3:.1f} us, gaps {(span - busy) / 1e3:.1f} us")
    print(f"{'t_us':>9s} {'dur_us':>8s} {'gap_us':>7s} {'wgs':>7s}  kernel")
    prev = seg[0][1]
    for name, s, e, g, w, gy, gz in seg:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*\)$", "", short).replace("void ", "")[:70]
        wgs = (g // max(w, 1)) * max(gy, 1) * max(gz, 1) if g else 0
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f} {(s - prev) / 1e3:7.2f} {wgs:7d}  {short}")
        prev = e


if __name__ == "__main__":
    main()
