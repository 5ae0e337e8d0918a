"""Fused clip-grad-norm + AdamW over the model's flat parameter buffer.

Drop-in for the optimizer the reference builds (trainer_base.py:130-198: transformers AdamW, lr 1e-4, eps 1e-6,
weight decay 0.01 except names containing "bias"/"LayerNorm.weight") followed by `clip_grad_norm_(params, 5)`
(vqacl.py:466-487), as two HIP kernels: one squared-norm reduction over the flat gradient buffer and one update
pass that also rewrites the bf16 shadow the GEMMs read (28 B/param + 2 B).  The clip coefficient stays on the
device, so the step never synchronises with the host.
"""
import torch

from . import _lib as L
from ._lib import check, lib, ptr, stream_ptr


def reference_param_groups(model, weight_decay=0.01):
    """The reference's grouping rule, by substring on the parameter NAME (trainer_base.py:148-161)."""
    no_decay = ["bias", "LayerNorm.weight"]
    named = list(model.named_parameters())
    return [
        {"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": weight_decay},
        {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0},
    ]


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, max_grad_norm=None,
                 hf_mode=True, overlap=False):
        """overlap=True: the update runs on a second HIP stream, bucket by bucket in the order the NEXT forward needs
        the weights (embeddings/norms, encoder bottom -> top, cross k/v, decoder bottom -> top), and the engine's forward
        phases wait per bucket -- the HBM-bound update (30 B/param) runs beside the encoder forward of the next step.  (On MI355X at B=80 this
        measured 2 % slower than the in-stream update -- the update's 6.7 GB stream evicts the forward's operands from
        L2/MALL -- so the bench leaves it off; it is kept for configurations with a longer, compute-bound forward.)
        Everything that goes through train_step/test_step/state_dict() is ordered automatically; code that reads
        parameters directly on another stream must call `model.sync_optimizer()` (or `optimizer.synchronize()`) first."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        model = getattr(model, "module", model)      # a DataParallelVLT5 wrapper: the engine-backed module carries the state
        self.model = model
        self.max_grad_norm = max_grad_norm
        self.hf_mode = int(hf_mode)
        flat = model.flat_params()
        self._m = torch.zeros_like(flat)
        self._v = torch.zeros_like(flat)
        self._t = 0
        dev = flat.device
        self._total_sq = torch.zeros(1, device=dev, dtype=torch.float32)
        base = flat.data_ptr()
        # the gradient-bearing region: everything up to the first never-used parameter (prototype_fc*)
        used_end = 0
        for name, (off, n, bucket, decay, used) in model._pinfo.items():
            if used:
                used_end = max(used_end, off + n)
        self._used_end = used_end
        self._partial = torch.empty(lib().vlt5_sqnorm_blocks(used_end) + 8, device=dev, dtype=torch.float32)
        # what the weight-gradient GEMMs' norm shares (VLT5._gnorm) do NOT cover: the last bucket (embeddings, norm weights, visual
        # embedding: scatter-added / column-summed gradients) and the two relative-position tables inside the layer buckets
        last = max(b for _, (off, n, b, decay, used) in model._pinfo.items() if used)
        lo = min(off for _, (off, n, b, decay, used) in model._pinfo.items() if used and b == last)
        rng = [(lo, used_end - lo)]
        for name, (off, n, b, decay, used) in model._pinfo.items():
            if used and b != last and name.endswith("relative_attention_bias.weight"):
                rng.append((off, n))
        import ctypes as C
        self._tail_off = (C.c_longlong * len(rng))(*[a for a, _ in rng])
        self._tail_n = (C.c_longlong * len(rng))(*[b for _, b in rng])
        self._tail_k = len(rng)
        # contiguous runs of equal hyper-parameters (alignment gaps hold zeros and stay zero under the update)
        spans = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                off = (p.data_ptr() - base) // 4
                if off < 0 or off + p.numel() > flat.numel():
                    raise L.Vlt5Error("FusedAdamW only handles parameters of the engine-backed model")
                if off >= used_end:
                    continue                         # never receives a gradient: skipped like a grad=None param
                spans.append((off, off + p.numel(), gi))
        spans.sort()
        runs = []
        for a, b, gi in spans:
            if runs and runs[-1][2] == gi and a - runs[-1][1] < 64:
                runs[-1][1] = b
            else:
                runs.append([a, b, gi])
        self._runs = runs
        # data parallel with bf16 buckets: read the reduced gradients straight from the staging buffer (no cast-back pass)
        dp = getattr(model, "dp", None)
        if dp is not None and dp.grad_dtype is torch.bfloat16 and flat.is_cuda:
            dp.defer_cast_back = True
        if dp is not None and dp.algo == "zero1" and overlap:
            raise L.Vlt5Error("FusedAdamW(overlap=True) and the sharded step of DataParallelVLT5(algo='zero1') both order the update "
                              "against the next forward through the per-bucket events: use one of them")
        if dp is not None and dp.algo == "zero1":
            # ZeRO-1 style step: this rank clips and updates only its chunk of every reduce-scattered slice, then the updated
            # chunks are all-gathered (parallel.py).  The all-gathers are ordered against the next forward by per-bucket events.
            dp.sharded_optimizer = True
            self._z_events = None
            self._zpartial = None
        self.overlap = bool(overlap)
        if self.overlap:
            # the same runs cut at bucket boundaries, grouped by bucket
            bounds = {}
            for name, (off, n, bucket, decay, used) in model._pinfo.items():
                if used and bucket >= 0:
                    lo, hi = bounds.get(bucket, (off, off + n))
                    bounds[bucket] = (min(lo, off), max(hi, off + n))
            self._nb = max(bounds) + 1
            self._bucket_runs = [[] for _ in range(self._nb)]
            for bkt, (lo, hi) in bounds.items():
                for a, b, gi in runs:
                    a2, b2 = max(a, lo), min(b, hi)
                    if a2 < b2:
                        self._bucket_runs[bkt].append((a2, b2, gi))
            covered = sum(b - a for rs in self._bucket_runs for a, b, _ in rs)
            if covered != sum(b - a for a, b, _ in runs):
                raise L.Vlt5Error("bucket ranges do not tile the parameter runs")
            self._side = torch.cuda.Stream(device=dev)
            self._events = [torch.cuda.Event() for _ in range(self._nb)]

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise L.Vlt5Error("closures are not supported")
        model = self.model
        flat, grad, bf16 = model._flat, model._flat_grad, model.flat_bf16()     # (flat_grads() would cast a deferred bucket back)
        anchor = model._params_by_name["shared.weight"]
        if anchor.grad is None:
            return None
        if anchor.grad.data_ptr() != model._gviews["shared.weight"].data_ptr():
            for name, p in model._params_by_name.items():           # grads were accumulated outside the flat buffer
                if p.grad is not None:
                    model._gviews[name].copy_(p.grad)
        self._t += 1
        dp = getattr(model, "dp", None)
        if dp is not None and dp.sharded_optimizer:
            # ZeRO-1: EVERY step is the sharded one -- the Adam moments (and, with gather_master=False, the f32 master) of the chunks
            # this rank does not own are stale from the first sharded step on, so a whole-buffer update must never run again.
            # Backward left the chunks reduce-scattered (shards_valid), or -- gradient accumulation, gradients read through
            # flat_grads() before the step -- every rank holds the fully reduced f32 gradients: its own chunks of them are used.
            return self._step_zero1(model, dp, flat, grad, bf16)
        g16 = dp._g16 if (dp is not None and dp.g16_valid and grad.data_ptr() == model._flat_grad.data_ptr()) else None
        gs = dp.grad_scale if g16 is not None else 1.0

        def update(runs, st, total):
            for a, b, gi in runs:
                g = self.param_groups[gi]
                tail = (L.vp(self._m.data_ptr() + 4 * a), L.vp(self._v.data_ptr() + 4 * a), L.vp(bf16.data_ptr() + 2 * a), b - a,
                        float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self._t,
                        ptr(total), float(self.max_grad_norm or 0.0), self.hf_mode, st)
                if g16 is None:
                    check(lib().vlt5_adamw_step(L.vp(flat.data_ptr() + 4 * a), L.vp(grad.data_ptr() + 4 * a), *tail), "vlt5_adamw_step")
                else:       # the reduced bf16 bucket as the all-reduce left it, times 1/world
                    check(lib().vlt5_adamw_step_g16(L.vp(flat.data_ptr() + 4 * a), L.vp(g16.data_ptr() + 2 * a), gs, *tail),
                          "vlt5_adamw_step_g16")

        def norm(st):
            if self.max_grad_norm is not None and self.max_grad_norm > 0:
                gn = getattr(model, "_gnorm", None)
                if (g16 is None and gn is not None and gn is not False and self._tail_k <= 4
                        and model._gnorm_version is not None and model._gnorm_version == grad._version):
                    # the weight-gradient GEMMs of this backward left their shares: sum them + the ranges no GEMM writes
                    check(lib().vlt5_gnorm_finish(ptr(gn), gn.numel(), ptr(grad), self._tail_off, self._tail_n, self._tail_k,
                                                  ptr(self._partial), ptr(self._total_sq), st), "vlt5_gnorm_finish")
                    return self._total_sq
                if g16 is None:
                    check(lib().vlt5_sqnorm(ptr(grad), self._used_end, ptr(self._partial), ptr(self._total_sq), 0, st), "vlt5_sqnorm")
                else:
                    check(lib().vlt5_sqnorm_g16(ptr(g16), gs, self._used_end, ptr(self._partial), ptr(self._total_sq), 0, st),
                          "vlt5_sqnorm_g16")
                return self._total_sq
            return None

        if not self.overlap:
            st = stream_ptr()
            update(self._runs, st, norm(st))
        else:
            model.sync_optimizer()                              # (a second step() without a forward in between)
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                st = stream_ptr()
                total = norm(st)
                for bkt in range(self._nb - 1, -1, -1):         # forward order: the last bucket holds embeddings + norms
                    update(self._bucket_runs[bkt], st, total)
                    self._events[bkt].record(self._side)
            model._opt_events = self._events
        if g16 is not None:
            dp.g16_valid = False            # consumed (the f32 gradient buffer keeps this rank's local gradients)
        model.external_bf16_sync = True
        model._bf16_version = flat._version
        return None

    def _step_zero1(self, model, dp, flat, grad, bf16):
        """Sharded step (DataParallelVLT5(algo="zero1")): the backward left chunk `rank` of every slice reduced; clip with the
        all-reduced squared norm of the chunks, update the chunks, all-gather them slice by slice in the order the next forward
        reads the weights."""
        import torch.distributed as dist
        scattered = dp.shards_valid
        slices = list(dp._slices_done) if scattered else dp.slice_plan()
        # the reduced bf16 chunks of this backward as the reduce-scatter left them (x 1/world), or f32: the f32 buffer holds this
        # rank's reduced chunks (f32 wire) / the fully reduced gradients (accumulation, flat_grads() before the step)
        use16 = scattered and dp.g16_valid and dp.grad_dtype is torch.bfloat16 and flat.is_cuda
        g16 = dp._g16 if use16 else None
        gs = dp.grad_scale if use16 else 1.0
        st = stream_ptr()
        if dp.master_ready is not None:
            torch.cuda.current_stream().wait_event(dp.master_ready)     # the previous step's master all-gather reads what this update writes
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        if clip:
            # block partials of every chunk side by side, ONE reduction at the end (a reduction per slice was a 10 us launch each)
            used = 0
            for a, b in slices:
                ca, cb = dp.chunk(a, b)
                cb = min(cb, self._used_end)
                if cb <= ca:
                    continue
                nblk = lib().vlt5_sqnorm_blocks(cb - ca)
                if self._zpartial is None:
                    self._zpartial = torch.empty(65536, device=flat.device, dtype=torch.float32)      # <= 2048 blocks per chunk
                if used + nblk > self._zpartial.numel():
                    raise L.Vlt5Error("too many gradient slices for the norm scratch")
                dst = L.vp(self._zpartial.data_ptr() + 4 * used)
                if use16:
                    check(lib().vlt5_sqnorm_g16(L.vp(g16.data_ptr() + 2 * ca), gs, cb - ca, dst, ptr(self._total_sq), 2, st), "vlt5_sqnorm_g16")
                else:
                    check(lib().vlt5_sqnorm(L.vp(grad.data_ptr() + 4 * ca), cb - ca, dst, ptr(self._total_sq), 2, st), "vlt5_sqnorm")
                used += nblk
            if used:
                check(lib().vlt5_gnorm_finish(ptr(self._zpartial), used, None, None, None, 0, ptr(self._partial), ptr(self._total_sq), st),
                      "vlt5_gnorm_finish")
            else:
                self._total_sq.zero_()
            dist.all_reduce(self._total_sq, group=dp.ctrl_group)
        total = self._total_sq if clip else None
        nb = len(dp.bucket_end)
        if self._z_events is None or len(self._z_events[0]) != len(slices):
            self._z_events = ([torch.cuda.Event() for _ in slices], [torch.cuda.Event() for _ in slices])
        done_main, done_ag = self._z_events
        bucket_events = [None] * nb
        # forward order: the last bucket (embeddings, norms) first, then descending offsets (encoder bottom -> decoder top)
        order = sorted(range(len(slices)), key=lambda i: -slices[i][0])
        cur = torch.cuda.current_stream()
        for j, i in enumerate(order):
            a, b = slices[i]
            ca, cb = dp.chunk(a, b)
            for ra, rb, gi in self._runs:
                x, y = max(ra, ca), min(rb, cb)
                if x >= y:
                    continue
                g = self.param_groups[gi]
                tail = (L.vp(self._m.data_ptr() + 4 * x), L.vp(self._v.data_ptr() + 4 * x), L.vp(bf16.data_ptr() + 2 * x), y - x,
                        float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self._t,
                        ptr(total), float(self.max_grad_norm or 0.0), self.hf_mode, st)
                if use16:
                    check(lib().vlt5_adamw_step_g16(L.vp(flat.data_ptr() + 4 * x), L.vp(g16.data_ptr() + 2 * x), gs, *tail),
                          "vlt5_adamw_step_g16")
                else:
                    check(lib().vlt5_adamw_step(L.vp(flat.data_ptr() + 4 * x), L.vp(grad.data_ptr() + 4 * x), *tail), "vlt5_adamw_step")
            done_main[j].record(cur)
            dp.allgather_updated(model, [(a, b)], done_main[j])
            done_ag[j].record(dp.comm_stream)
            for bkt in range(nb):
                if a <= dp.bucket_start[bkt] and dp.bucket_end[bkt] <= b:
                    bucket_events[bkt] = done_ag[j]
        assert all(e is not None for e in bucket_events), "slices do not cover the buckets"
        model._opt_events = bucket_events
        dp._param_slices = slices
        dp._slices_done = []
        dp.shards_valid = False
        dp.g16_valid = False
        if dp.gather_master:
            dp.gather_master_async(model, [slices[i] for i in order])
            dp.params_sharded = False
        else:
            dp.params_sharded = dp.world > 1
        model.external_bf16_sync = True
        model._bf16_version = flat._version
        return None

    def synchronize(self):
        """Order the current stream after an overlapped update."""
        self.model.sync_optimizer()

    def grad_norm(self):
        """L2 norm of the last step's (pre-clip) gradients; reading it synchronises."""
        self.model.sync_optimizer()
        return float(self._total_sq.sqrt())
