#!/usr/bin/env python3
"""Phase timeline of the decode projection kernel (csrc/decode.hip, a -DDECLIN_TIMELINE build: VLT5_LIB=vqacl_amd/libvlt5_tl.so): shader
clock stamps of wave 0 of every workgroup over one greedy-decoding step of VL-T5-base at B = 80 -- kernel entry, loads issued, norm
weights staged, MFMAs done, partial tiles met, stores issued, stores landed.  Prints per launch of one decoder layer the median over the
workgroups of each phase, and the distance from the last stamp of a launch to the first stamp of the next (the launch boundary)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_cpu import Cfg, synthetic_batch  # noqa: E402  (the synthetic-input recipe only)
from vqacl_amd import VLT5Config, VLT5VQA  # noqa: E402
from vqacl_amd._lib import LIB_PATH  # noqa: E402

B = 80
dev = torch.device("cuda")
torch.manual_seed(1)
model = VLT5VQA(VLT5Config(dropout_rate=0.1), device=dev)
model.eval()
model.decode_graph = False          # every launch gets its own stamp block from the host side of the launch: enqueue the steps, do not replay them
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic_batch(Cfg(), B=B, L=20, V=36, T=5, seed=3, task_id=0).items()}
fb = (batch["vis_feats"], batch["boxes"])
raw = C.CDLL(LIB_PATH)
assert hasattr(raw, "vlt5_declin_timeline"), "needs a -DDECLIN_TIMELINE build (bash tools/build_variant.sh tl -DDECLIN_TIMELINE)"
raw.vlt5_declin_timeline.argtypes = [C.c_void_p, C.c_longlong]
for _ in range(2):
    model.greedy_generate(batch["input_ids"], fb, max_length=8, eos_token_id=-1)
STEPS, PER_STEP, WGS = 6, 12 * 6, 2520          # (the vocabulary projection runs its own kernels: no stamp block)
buf = torch.zeros(STEPS * PER_STEP, WGS * 8, dtype=torch.int64, device=dev)
raw.vlt5_declin_timeline(C.c_void_p(buf.data_ptr()), WGS * 8)
model.greedy_generate(batch["input_ids"], fb, max_length=STEPS + 1, eos_token_id=-1)
torch.cuda.synchronize()
raw.vlt5_declin_timeline(None, 0)
t = buf.cpu().view(STEPS, PER_STEP, WGS, 8)
names = ["norm->q|k|v", "self o + res", "norm->cross q", "cross o + res", "norm->wi relu", "wo + res"]
step = 4
print(f"# decode step {step}, decoder layers 5 and 6 (launches {5 * 6}..{7 * 6 - 1} of the step's declin launches); cycles of the shader clock, median over workgroups")
print(f"{'launch':16s} {'wgs':>5s} {'issue':>7s} {'w staged':>9s} {'mfma done':>10s} {'met':>7s} {'stores out':>11s} {'landed':>8s} {'total':>7s} | first-in to last-out (all wgs) | gap to next launch's first stamp")
prev_end = None
for li in range(5 * 6, 7 * 6):
    x = t[step, li][0::8]                      # workgroups of XCD 0 only (block b runs on XCD b % 8; the clocks of two XCDs need not agree)
    live = x[:, 0] > 0
    x = x[live]
    n = x.shape[0]
    s0 = x[:, 0]
    def med(i):
        v = x[:, i]
        return int((v[v > 0] - s0[v > 0]).median()) if bool((v > 0).any()) else 0
    first, last = int(s0.min()), int(x[:, 6].max())
    nxt = t[step, li + 1][0::8]
    nfirst = int(nxt[:, 0][nxt[:, 0] > 0].min())
    print(f"{names[li % 6]:16s} {n:5d} {med(1):7d} {med(2):9d} {med(3):10d} {med(4):7d} {med(5):11d} {med(6):8d} {int((x[:, 6] - s0).median()):7d} | {last - first:8d} | {nfirst - last:8d}")
