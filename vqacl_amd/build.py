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


def source_hash():
    """sha256 (16 hex digits) over the HIP sources, their headers and the C ABI header: the identity of the build a profile was taken
    with (tools/rocpd_stats.py writes it into the kernel-stats file, bench.py only quotes a trace whose hash equals the tree's)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith((".hip", ".h")) or f == "Makefile")
    files.append(os.path.join(os.path.dirname(HERE), "include", "vlt5_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
