#!/usr/bin/env python3
"""Build the HIP library of ANOTHER git revision next to the current one, for same-box A/B runs (box-to-box spread is ~2 %, larger
than most kernel changes): python tools/ab_build.py <git-ref>  ->  vqacl_amd/libvlt5_<ref>.so
Then on the GPU box:   bash tools/ab_run.sh vqacl_amd/libvlt5_<ref>.so vqacl_amd/libvlt5_hip.so
(`VLT5_LIB` selects the library, vqacl_amd/_lib.py; only kernel-side changes can be compared this way -- the Python side is shared,
so the two revisions must agree on the C ABI.)"""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = sys.argv[1]
tag = re.sub(r"[^A-Za-z0-9_.-]", "_", ref)
work = os.path.join(ROOT, "build", "ab_" + tag)
shutil.rmtree(work, ignore_errors=True)
os.makedirs(os.path.join(work, "vqacl_amd", "csrc"))
os.makedirs(os.path.join(work, "include"))
files = subprocess.check_output(["git", "-C", ROOT, "ls-tree", "-r", "--name-only", ref, "vqacl_amd/csrc", "include"], text=True).split()
for f in files:
    data = subprocess.check_output(["git", "-C", ROOT, "show", f"{ref}:{f}"])
    with open(os.path.join(work, f), "wb") as fh:
        fh.write(data)
subprocess.check_call(["make", "-C", os.path.join(work, "vqacl_amd", "csrc"), "-j8"])
out = os.path.join(ROOT, "vqacl_amd", f"libvlt5_{tag}.so")
shutil.copy(os.path.join(work, "vqacl_amd", "libvlt5_hip.so"), out)
shutil.rmtree(work, ignore_errors=True)         # (build/ travels to the GPU box with every gpurun call)
print(out)
