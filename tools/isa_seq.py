#!/usr/bin/env python3
"""Order of memory / matrix / wait instructions of one kernel in a hipcc -save-temps .s file (L = global load, M = MFMA, S = global
store, D/d = LDS read/write, wN = s_waitcnt vmcnt(N), | = barrier).  usage: python tools/isa_seq.py file.s <substring of the mangled name>"""
import re
import sys

s = open(sys.argv[1]).read()
for key in sys.argv[2:]:
    m = re.search(r"\n(_Z\w*" + re.escape(key) + r"\w*): +; @", s)
    if not m:
        print(key, "not found")
        continue
    i = m.end()
    j = s.index("s_endpgm", i)
    seq = []
    for line in s[i:j].splitlines():
        t = line.strip()
        if t.startswith("global_load") or t.startswith("buffer_load"):
            seq.append("L")
        elif t.startswith("global_store"):
            seq.append("S")
        elif t.startswith("v_mfma"):
            seq.append("M")
        elif t.startswith("ds_read") or t.startswith("ds_load"):
            seq.append("D")
        elif t.startswith("ds_write") or t.startswith("ds_store"):
            seq.append("d")
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            seq.append("w" + re.search(r"vmcnt\((\d+)\)", t).group(1))
        elif t.startswith("s_barrier"):
            seq.append("|")
        elif t.startswith("s_cbranch"):
            seq.append("br")
    print(m.group(1), "\n  ", " ".join(seq))
