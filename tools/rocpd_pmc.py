#!/usr/bin/env python3
"""Per-kernel average of a rocprofv3 PMC counter from a rocpd .db (one counter per pass, as the MI355X guide prescribes).
usage: python tools/rocpd_pmc.py fetch.db write.db > profiles/..._pmc_hbm_traffic.txt
FETCH_SIZE / WRITE_SIZE are kilobytes; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for wide coalesced reads
(MI355X_MICROARCH.md, HBM): the 'corrected' column doubles it."""
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqacl_amd.build import source_hash  # noqa: E402


def per_kernel(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    namecol = "display_name" if "display_name" in kcols else "kernel_name"
    q = f"""select s.{namecol}, count(*), sum(p.value), i.name
            from rocpd_pmc_event p join rocpd_kernel_dispatch d on p.event_id = d.event_id
            join rocpd_info_kernel_symbol s on d.kernel_id = s.id join rocpd_info_pmc i on p.pmc_id = i.id
            group by s.{namecol}, i.name order by 3 desc"""
    out = {}
    for name, n, tot, cname in cur.execute(q):
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*\)$", "", short)
        out[short] = (n, tot, cname)
    return out


def main():
    f = per_kernel(sys.argv[1])
    w = per_kernel(sys.argv[2])
    rows = [dict(source_sha16=source_hash())]          # the build these passes were taken with (bench.py: traffic_build_matches)
    print(f"# source_sha16 {source_hash()}")
    print(f"{'kernel':60s} {'calls':>6s} {'FETCH_KB/launch':>16s} {'x2 (gfx950)':>12s} {'WRITE_KB/launch':>16s} {'HBM MB/launch':>14s}")
    for k in sorted(f, key=lambda k: -f[k][1]):
        n, tot, _ = f[k]
        wn, wtot, _ = w.get(k, (1, 0.0, ""))
        fk, wk = tot / n, wtot / max(wn, 1)
        hbm = (2 * fk + wk) / 1024.0
        rows.append(dict(kernel=k, calls=n, fetch_kb=fk, fetch_kb_corrected=2 * fk, write_kb=wk, hbm_mb=hbm))
        print(f"{k[:60]:60s} {n:6d} {fk:16.1f} {2 * fk:12.1f} {wk:16.1f} {hbm:14.2f}")
    if len(sys.argv) > 3:
        json.dump(rows, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
