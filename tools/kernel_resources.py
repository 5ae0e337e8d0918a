#!/usr/bin/env python3
"""Per-kernel register / occupancy table of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage, no GPU needed).
usage: python tools/kernel_resources.py vqacl_amd/csrc/gemm.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{ROOT}/vqacl_amd/csrc",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in err.splitlines():
    m = re.search(r"remark: +([A-Za-z ]+?(?: \[[^\]]*\])?): *(.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:70]}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" [")[0]] = v
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'occ':>4s} {'sgprspill':>9s} {'vspill':>6s} {'scratch':>7s}")
for r in rows:
    print(f"{r['name']:70s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('Occupancy','?'):>4s} {r.get('SGPRs Spill','?'):>9s} "
          f"{r.get('VGPRs Spill','?'):>6s} {r.get('ScratchSize','?'):>7s}")
