"""Builds vqacl_amd/libvlt5_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose=False, jobs=8):
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), f"-j{jobs}"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise RuntimeError("building libvlt5_hip.so failed")
    return os.path.join(HERE, "libvlt5_hip.so")
