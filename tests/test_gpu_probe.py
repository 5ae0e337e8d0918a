"""Hardware-semantics probes the kernels rely on (gfx950)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_ds_read_b64_tr_b16_semantics():
    """out[lane l][j] = X[16*(l>>4) + 4*j + ((l&15)>>2)][l&3], X[i][e] = the e-th 16-bit element at lane i's own address.
    The k-major GEMM operands are read with this instruction (csrc/gemm.hip)."""
    from vqacl_amd import _lib
    raw = C.CDLL(_lib.LIB_PATH)
    dev = torch.device("cuda")
    n = 4096
    src = torch.arange(n, dtype=torch.int16, device=dev)
    g = torch.Generator().manual_seed(0)
    addr = (torch.randint(0, (n - 4) // 4, (64,), generator=g) * 4).to(torch.int32).to(dev)     # 8-byte aligned, arbitrary
    out = torch.zeros(256, dtype=torch.int16, device=dev)
    raw.vlt5dbg_tr_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    rc = raw.vlt5dbg_tr_read(src.data_ptr(), out.data_ptr(), addr.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    got = out.cpu().view(64, 4)
    a = addr.cpu()
    exp = torch.zeros(64, 4, dtype=torch.int16)
    for l in range(64):
        for j in range(4):
            src_lane = 16 * (l >> 4) + 4 * j + ((l & 15) >> 2)
            exp[l, j] = int(a[src_lane]) + (l & 3)
    if not torch.equal(got, exp):
        # print the observed mapping to help re-derive it
        rows = []
        for l in range(64):
            rows.append([(int((a == (int(v) // 4) * 4).nonzero()[0]) if ((a == (int(v) // 4) * 4).any()) else -1, int(v) % 4) for v in got[l]])
        print("observed (source lane, element) per output lane:", rows)
    assert torch.equal(got, exp)
