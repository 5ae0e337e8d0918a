#!/usr/bin/env python3
"""EXPERIMENT: the fused RMS-norm + q|k|v projection per (sample, head group) (vqacl_amd/csrc/experiments/ln_qkv_probe.hip) against the two
launches it would replace (vlt5_layernorm_fwd + vlt5_gemm_bf16): correctness and HIP-event time at the encoder's shape.
build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I include -I vqacl_amd/csrc vqacl_amd/csrc/experiments/ln_qkv_probe.hip -o vqacl_amd/libxp_probe.so"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vqacl_amd import ops  # noqa: E402
from vqacl_amd._lib import lib, ptr, stream_ptr  # noqa: E402

xp = C.CDLL(os.path.join(ROOT, "vqacl_amd", "libxp_probe.so"))
vp = C.c_void_p
xp.xp_ln_qkv.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]
xp.xp_ln_qkv.restype = C.c_int
dev = torch.device("cuda:0")
BF = torch.bfloat16
B, S, d, N = 80, 56, 768, 2304
x = torch.randn(B * S, d, device=dev)
lnw = (1 + 0.1 * torch.randn(d, device=dev))
W = (torch.randn(N, d, device=dev) * 0.05).to(BF)
out = torch.zeros(B * S, N, device=dev, dtype=BF)
rstd = torch.zeros(B * S, device=dev)


def fused():
    rc = xp.xp_ln_qkv(ptr(x), ptr(lnw), ptr(W), ptr(out), ptr(rstd), B, S, 3, N // 3, N, 1e-6, stream_ptr())
    assert rc == 0, rc


yb = torch.empty(B * S, d, device=dev, dtype=BF)
rs2 = torch.empty(B * S, device=dev)
ref = torch.empty(B * S, N, device=dev, dtype=BF)
g, _, keep = ops.gemm_desc(yb, W, B * S, N, d, out=ref)


def separate():
    lib().vlt5_layernorm_fwd(ptr(x), ptr(lnw), ptr(yb), None, ptr(rs2), B * S, d, 1e-6, 0.0, 0, 0, 0, stream_ptr())
    lib().vlt5_gemm_bf16(C.byref(g), stream_ptr())


fused()
separate()
torch.cuda.synchronize()
err = float((out.float() - ref.float()).abs().max() / ref.float().abs().max())
print(f"max rel err vs the two launches: {err:.3e}   rstd err {float((rstd - rs2).abs().max()):.2e}")


def t(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"fused norm + qkv (240 workgroups): {t(fused):.1f} us     norm launch + GEMM launch: {t(separate):.1f} us")
