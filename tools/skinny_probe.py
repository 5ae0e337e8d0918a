#!/usr/bin/env python3
"""Row-panel (skinny) GEMM against the tiled kernel on the decoder's shapes: correctness vs an f32 torch reference, graph-replayed
timing warm (one weight matrix) and cold-ish (rotating through `--rot` weight matrices, > Infinity Cache).
usage: make -C vqacl_amd/csrc exp; VLT5_LIB=vqacl_amd/libvlt5_exp.so python tools/skinny_probe.py ["M N K ln relu resid" ...]"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_sweep import timed_graph
from tools.experiments import skinny_py as ops_sk
from vqacl_amd import ops
from vqacl_amd._lib import lib, stream_ptr
dev = torch.device("cuda")
BF = torch.bfloat16
specs = [a for a in sys.argv[1:] if not a.startswith("--")] or [
    "400 2304 768 1 0 0", "400 768 768 0 0 1", "400 768 768 1 0 0", "400 3072 768 1 1 0", "400 768 3072 0 0 1",
    "4480 768 768 0 0 1", "4480 768 3072 0 0 1", "4480 2304 768 1 0 0", "4480 3072 768 1 1 0"]
ROT = 48
torch.manual_seed(0)
for spec in specs:
    M, N, K, ln, relu, res = [int(v) for v in spec.split()]
    X = torch.randn(M, K, device=dev) * 3
    lw = torch.rand(K, device=dev) + 0.5
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(BF) for _ in range(ROT if M <= 512 else 8)]
    R = torch.randn(M, N, device=dev) if res else None
    if ln:
        xn_ref = (X * torch.rsqrt((X * X).mean(1, keepdim=True) + 1e-6) * lw).to(BF)
        A = None
    else:
        A = X.to(BF)
        xn_ref = A
    ref = xn_ref.float() @ Ws[0].float().t()
    if relu:
        ref = ref.relu()
    if res:
        ref = ref + R
    for rows, cols in ((0, 0), (32, 64), (32, 128), (32, 192), (16, 64), (16, 128), (16, 192)):
        if rows == 32 and K * 64 > 64 * 1024 * 2 and K * 64 > 144 * 1024:
            continue
        kw = dict(A=A, ln_x=X if ln else None, ln_w=lw if ln else None, relu=bool(relu), resid=R, out_f32=bool(res), panel_rows=rows,
                  chunk_cols=cols)
        try:
            out = ops_sk.skinny_gemm(Ws[0], M, N, K, **kw)
        except Exception as e:
            print(f"  {spec} panel {rows} x {cols}: {e}")
            continue
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        descs = [ops_sk.skinny_desc(W, M, N, K, out=out, **kw) for W in Ws]
        fn = ops_sk.bind().vlt5_skinny_gemm
        sp = stream_ptr()
        warm = timed_graph(lambda: fn(C.byref(descs[0][0]), stream_ptr()))
        it = [0]

        def rot():
            fn(C.byref(descs[it[0] % len(descs)][0]), stream_ptr())
            it[0] += 1
        cold = timed_graph(rot, reps=len(descs))
        print(f"M={M} N={N} K={K} ln={ln} relu={relu} res={res} panel {rows:2d}x{cols:3d}: err {err:.1e}  warm {warm:6.2f} us  rot {cold:6.2f} us "
              f"({2.0 * M * N * K / cold / 1e6:6.1f} TF)", flush=True)
    # the tiled kernel on the same product (bf16 A given; the norm is a separate launch there)
    A2 = xn_ref.contiguous()
    o2 = torch.empty(M, N, device=dev, dtype=torch.float32 if res else BF)
    gd = [ops.gemm_desc(A2, W, M, N, K, out=o2, relu=bool(relu), resid=R) for W in Ws]
    fn2 = lib().vlt5_gemm_bf16
    warm = timed_graph(lambda: fn2(C.byref(gd[0][0]), stream_ptr()))
    it = [0]

    def rot2():
        fn2(C.byref(gd[it[0] % len(gd)][0]), stream_ptr())
        it[0] += 1
    cold = timed_graph(rot2, reps=len(gd))
    err = float((o2.float() - ref).abs().max() / ref.abs().max())
    print(f"M={M} N={N} K={K} tiled kernel (auto, no norm):        err {err:.1e}  warm {warm:6.2f} us  rot {cold:6.2f} us ({2.0 * M * N * K / cold / 1e6:6.1f} TF)", flush=True)
    if ln:
        us = timed_graph(lambda: ops.layernorm_fwd(X, lw))
        print(f"   + separate ln_fwd launch {us:6.2f} us")
