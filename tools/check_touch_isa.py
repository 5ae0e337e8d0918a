#!/usr/bin/env python3
"""The panel-touch requests of the GEMM kernels (gemm_kernel.h GEMM_TOUCH_B, enc_attn.hip) are plain loads whose destination register
is written ASYNCHRONOUSLY, when the load lands.  The compiler does not know that: if it copies the destination and reuses the
register, the late write clobbers whatever lives there (round 6: a memory fault through a clobbered address register).  This scans the
generated ISA of every kernel for the pattern: after an inline-asm `global_load_dword vN, ..., off`, no instruction may WRITE vN before the
next counted wait on the vector-memory counter (`s_waitcnt vmcnt(...)`), by which time the request -- the oldest in the queue -- has
landed.  No GPU needed.      python tools/check_touch_isa.py            (exit code 1 on a finding)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vqacl_amd", "csrc")
# translation units that can contain such a request: sources with an inline-asm `global_load_dword %0` themselves, or including a header that has one
PAT = "global_load_dword %0"
_hdrs = [h for h in os.listdir(CSRC) if h.endswith(".h") and PAT in open(os.path.join(CSRC, h)).read()]
FILES = [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and
         (PAT in open(os.path.join(CSRC, f)).read() or any(f'#include "{h}"' in open(os.path.join(CSRC, f)).read() for h in _hdrs))]
dst_re = re.compile(r"^\s+(\S+)\s+(v\d+|v\[\d+:\d+\])\b")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(tok[1:])}


def isa_of(f):
    return subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{CSRC}", "-S", "--cuda-device-only",
                           os.path.join(CSRC, f), "-o", "-"] + sys.argv[1:], capture_output=True, text=True).stdout.splitlines()


from concurrent.futures import ThreadPoolExecutor  # noqa: E402
with ThreadPoolExecutor(max_workers=8) as pool:
    listings = dict(zip(FILES, pool.map(isa_of, FILES)))
bad = total = 0
for f in FILES:
    asm = listings[f]
    kernel, in_asm = None, False
    pending = {}                                     # register -> line number of its request
    for n, line in enumerate(asm, 1):
        if re.match(r"^[_A-Za-z][\w$.]*:\s*(;.*)?$", line) and not line.startswith(".L"):
            kernel, pending = line.split(":")[0], {}
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        m = re.match(r"\s+global_load_dword (v\d+), v\[\d+:\d+\], off\s*$", line)
        if in_asm and m:
            pending[int(m.group(1)[1:])] = n
            total += 1
            continue
        if "s_waitcnt" in line and "vmcnt" in line:
            pending = {}
            continue
        if line.strip().startswith("s_endpgm"):
            pending = {}
            continue
        m = dst_re.match(line)
        if m and pending and not m.group(1).startswith(("global_store", "buffer_store", "ds_write", "ds_store", "scratch_store", "v_cmp", "s_")):
            hit = regs(m.group(2)) & set(pending)
            for r in hit:
                bad += 1
                print(f"{f}: {str(kernel)[:90]}\n   line {n}: `{line.strip()}` writes v{r}, requested at line {pending[r]} and not waited for yet")
                pending.pop(r)
print(f"{total} touch requests in {len(FILES)} files, {bad} written before a counted wait")
sys.exit(1 if bad else 0)
