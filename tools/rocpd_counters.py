#!/usr/bin/env python3
"""Per-kernel average (per dispatch) of every PMC counter found in the given rocpd .db files (one rocprofv3 --pmc pass each).
usage: python tools/rocpd_counters.py pass1.db pass2.db ... [--match gemm]"""
import re
import sqlite3
import sys


def main():
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
    paths = [a for a in sys.argv[1:] if a.endswith(".db")]
    table = {}
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        kcols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
        namecol = "display_name" if "display_name" in kcols else "kernel_name"
        q = f"""select s.{namecol}, i.name, count(distinct d.id), sum(p.value)
                from rocpd_pmc_event p join rocpd_kernel_dispatch d on p.event_id = d.event_id
                join rocpd_info_kernel_symbol s on d.kernel_id = s.id join rocpd_info_pmc i on p.pmc_id = i.id
                group by s.{namecol}, i.name"""
        for name, cname, n, tot in cur.execute(q):
            short = re.sub(r"\(anonymous namespace\)::", "", name)
            short = re.sub(r"\(.*\)$", "", short)
            if match and match not in short:
                continue
            table.setdefault(short, {})[cname] = tot / max(n, 1)
    for k, row in table.items():
        print(f"== {k}")
        for c in sorted(row):
            print(f"   {c:40s} {row[c]:16.1f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in row and "GRBM_GUI_ACTIVE" in row:
            # GRBM_GUI_ACTIVE is reported once per XCD (8 on MI355X) and summed here; 1024 SIMDs on the chip
            cyc = row["GRBM_GUI_ACTIVE"] / 8.0
            print(f"   {'-> cycles per launch':40s} {cyc:16.1f}")
            print(f"   {'-> MfmaUtil % (MFMA busy / (cycles*1024))':40s} {100.0 * row['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):16.2f}")


if __name__ == "__main__":
    main()
