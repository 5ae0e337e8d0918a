"""EXPERIMENT, not product (vqacl_amd/placement.py of tools/experiments/stream_placement.patch; measured in round 6, no reliable gain: profiles/r06_i_*).

Where the optimizer pass's five streams lie in HBM (round 6).

The clip + AdamW pass (`csrc/optim.hip`; reference: transformers AdamW + clip_grad_norm_, trainer_base.py:130-198, vqacl.py:466-487) reads
the f32 master, the gradient and both moments and writes the master, both moments and the bf16 shadow -- 30 B per parameter, eight
streams advancing in lockstep over buffers of 0.9 GB each (VL-T5-base).  Its time depends on WHERE those buffers lie relative to each
other: with unchanged code the same nine launches take 1.13-1.46 ms per step over the committed traces of rounds 2-6, eight independent
allocation sets in ONE process on ONE box give 1.19 ... 1.43 ms (each stable to 0.5 %), and carved out of one allocation the pass moves
from 1.46 to 1.27 ms with the byte offset between the streams (`tools/adamw_placement_probe.py`, `tools/adamw_pad_probe.py`,
`profiles/r06_h_*`).  There is no closed form for the good offsets (they differ between boxes), so the placement is MEASURED: the model
times the real AdamW kernel on sixteen candidate placements of master, gradient, moments and shadow -- one arena at twelve strides
between the streams, four sets of plain separate allocations (on some boxes one of those beats every stride) -- ~6 ms each, once per
model construction, and keeps the fastest.

Numerics are not involved: the buffers hold what they always held, only their addresses are chosen.  Host logic without a GPU (CPU
models of the tests) and small models get plain separate allocations.
"""
import os

import torch

MIB = 1 << 20
# byte offsets added to the streams' natural stride (4 * n bytes); the arena carries 4 * max(PADS) of slack (288 GB of HBM)
PADS = tuple(k * 24 * MIB for k in range(12))      # 0 ... 264 MiB: no structure was found in the good offsets, so a dozen are simply tried
MIN_ELEMS = 32 * 1000 * 1000          # below this the pass is microseconds: not worth a measurement


def _alias(storage, byte_off, n, dtype, device):
    """A tensor of n elements on `storage` at `byte_off` that is NOT an autograd view of anything: its own version counter (the model
    tells "the master was written by torch" from `_flat._version`; views of one arena would share a single counter)."""
    t = torch.empty(0, dtype=dtype, device=device)
    esz = t.element_size()
    assert byte_off % esz == 0
    t.set_(storage, byte_off // esz, (n,))
    return t


def carve(arena, n, pad):
    """(master, grad, m, v, shadow) of n elements each out of the uint8 arena, consecutive f32 streams 4n + pad bytes apart."""
    st, dev = arena.untyped_storage(), arena.device
    base = arena.storage_offset()
    stride = 4 * n + pad
    f32 = [_alias(st, base + k * stride, n, torch.float32, dev) for k in range(4)]
    return f32[0], f32[1], f32[2], f32[3], _alias(st, base + 4 * stride, n, torch.bfloat16, dev)


def _time_pass(bufs, n, reps=3):
    from ._lib import lib, ptr, stream_ptr
    p, g, m, v, pb = bufs
    tot = torch.ones(1, device=p.device)
    step = [0]

    def run():
        step[0] += 1
        rc = lib().vlt5_adamw_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(pb), n, 1e-4, 0.9, 0.999, 1e-6, 0.01, step[0], ptr(tot), 5.0, 1,
                                   stream_ptr())
        if rc:
            raise RuntimeError(f"vlt5_adamw_step -> {rc}")
    run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


N_SEPARATE = 4                         # independent allocation sets tried beside the arena strides


def place_streams(n, device, pads=None, log=None):
    """The five flat buffers of a model with n parameters (n a multiple of 64 elements): (master, grad, shadow, m, v, info).  m / v are
    None when nothing was measured (CPU, small model, VQACL_PLACEMENT=0): the optimizer then allocates its own moments.  Otherwise
    the candidates -- the arena at each stride of PADS and N_SEPARATE sets of five plain allocations (what the model did until round
    6; on some boxes one of those beats every arena stride) -- are all allocated, the real AdamW kernel is timed on each, the fastest
    is kept and the rest freed.  info: what was measured."""
    dev = torch.device(device)
    mode = os.environ.get("VQACL_PLACEMENT", "1")
    if dev.type != "cuda" or n < MIN_ELEMS or mode == "0":
        return (torch.zeros(n, device=dev, dtype=torch.float32), torch.zeros(n, device=dev, dtype=torch.float32),
                torch.zeros(n, device=dev, dtype=torch.bfloat16), None, None, {"placement": "separate allocations, not measured"})
    pads = tuple(PADS if pads is None else pads)
    n_sep = N_SEPARATE
    if mode not in ("0", "1"):                     # VQACL_PLACEMENT=<MiB>: the arena at that stride, no measurement (A/B runs)
        pads, n_sep = (int(mode) * MIB,), 0
    nbytes = 4 * (4 * n + max(pads)) + 2 * n + 4096
    with torch.cuda.device(dev):
        arena = torch.zeros(nbytes, device=dev, dtype=torch.uint8)
        sets = [[torch.zeros(n, device=dev, dtype=torch.float32) for _ in range(4)] + [torch.zeros(n, device=dev, dtype=torch.bfloat16)]
                for _ in range(n_sep)]
        timings, cands = {}, []
        for pad in pads:
            cands.append((f"arena+{pad // MIB}MiB", lambda pad=pad: carve(arena, n, pad)))
        for i, bufs in enumerate(sets):
            cands.append((f"separate#{i}", lambda bufs=bufs: tuple(bufs)))
        best = cands[0][0]
        if len(cands) > 1:
            for name, make in cands:
                bufs = make()
                bufs[1].fill_(1e-3)                # (a gradient that moves the moments: the pass does its real arithmetic)
                timings[name] = round(_time_pass(bufs, n), 1)
                torch.cuda.synchronize(dev)
            best = min(timings, key=timings.get)
        master, grad, m, v, shadow = dict(cands)[best]()
        if best.startswith("arena"):
            arena.zero_()                          # the trial passes wrote masters and moments
            master._vlt5_arena = arena
        else:
            for t in (master, grad, m, v, shadow):
                t.zero_()
        del sets, cands, arena
        torch.cuda.empty_cache()                   # the losing candidates go back to the driver, not into the caching allocator's pool
    info = {"placement": best, "adamw_us_by_candidate": timings}
    if log:
        log(info)
    return master, grad, shadow, m, v, info
