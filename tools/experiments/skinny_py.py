"""ctypes side of the row-panel GEMM experiment (tools/experiments/skinny.hip; built by `make -C vqacl_amd/csrc exp` into
vqacl_amd/libvlt5_exp.so, which also contains the whole library: run the probes with VLT5_LIB pointing at it)."""
import ctypes as C

import torch

from vqacl_amd import _lib as L
from vqacl_amd._lib import check, lib, ptr, stream_ptr
from vqacl_amd.ops import _need, BF16

c_i, c_f, c_u32, vp = L.c_i, L.c_f, L.c_u32, L.vp


class SkinnyDesc(C.Structure):
    _fields_ = [("A", vp), ("lda", c_i), ("ln_x", vp), ("ldx", c_i), ("ln_w", vp), ("eps", c_f), ("rstd_out", vp), ("xn_out_bf16", vp),
                ("W", vp), ("ldw", c_i), ("C", vp), ("ldc", c_i), ("out_f32", c_i), ("M", c_i), ("N", c_i), ("K", c_i), ("alpha", c_f),
                ("relu", c_i), ("drop_p", c_f), ("drop_seed", c_u32), ("resid", vp), ("ldr", c_i), ("panel_rows", c_i), ("chunk_cols", c_i)]



def bind():
    l = lib()
    l.vlt5_skinny_gemm.restype, l.vlt5_skinny_gemm.argtypes = c_i, [C.POINTER(SkinnyDesc), vp]
    l.vlt5_skinny_ok.restype, l.vlt5_skinny_ok.argtypes = c_i, [c_i, c_i, c_i, c_i]
    return l


def skinny_desc(W, M, N, K, *, A=None, ln_x=None, ln_w=None, eps=1e-6, out=None, out_f32=False, alpha=1.0, relu=False, resid=None,
                drop_p=0.0, drop_seed=0, rstd_out=None, xn_out=None, panel_rows=0, chunk_cols=0):
    """The filled vlt5_skinny_desc (row-panel GEMM, optional RMS-norm prologue): (desc, out, keep-alive)."""
    _need(W, BF16)
    if out is None:
        out = torch.empty(M, N, device=W.device, dtype=torch.float32 if out_f32 else BF16)
    g = SkinnyDesc()
    if A is not None:
        g.A, g.lda = ptr(_need(A, BF16)), A.stride(0)
    if ln_x is not None:
        g.ln_x, g.ldx, g.ln_w, g.eps = ptr(_need(ln_x, torch.float32)), ln_x.stride(0), ptr(ln_w), eps
        g.rstd_out, g.xn_out_bf16 = ptr(rstd_out), ptr(xn_out)
    g.W, g.ldw = ptr(W), W.stride(0)
    g.C, g.ldc, g.out_f32 = ptr(out), out.stride(0), int(out.dtype == torch.float32)
    g.M, g.N, g.K, g.alpha = M, N, K, alpha
    g.relu, g.drop_p, g.drop_seed = int(relu), drop_p, drop_seed
    g.resid, g.ldr = ptr(resid), (resid.stride(0) if resid is not None else 0)
    g.panel_rows, g.chunk_cols = panel_rows, chunk_cols
    return g, out, (A, ln_x, ln_w, W, resid, rstd_out, xn_out)


def skinny_gemm(W, M, N, K, **kw):
    g, out, _keep = skinny_desc(W, M, N, K, **kw)
    check(bind().vlt5_skinny_gemm(C.byref(g), stream_ptr()), "vlt5_skinny_gemm")
    return out


