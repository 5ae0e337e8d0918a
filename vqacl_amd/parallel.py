"""Data parallelism for the engine-backed VLT5: one process per GPU, RCCL collectives over xGMI.

The reference wraps the model in DDP but calls `model.module.train_step`, which bypasses DDP's reducer, so its
ranks never synchronise gradients.  This wrapper provides the synchronisation the north star asks
for: because the gradients live in ONE flat buffer laid out in backward-completion order, a bucket is just a
slice -- no flatten/unflatten copies.  The engine records a HIP event when each layer's gradients are complete;
the comm stream waits on the event and reduces that slice while the compute stream continues with the next
layer (overlap with backward).  xGMI is point-to-point, so few, large collectives are preferred: consecutive
layer buckets are merged up to `bucket_mb`.

Gradient exchange (`algo`):
  * "allreduce": one `all_reduce` per merged slice.
  * "rs_ag":     `reduce_scatter_tensor` in backward (rank r ends up with the reduced chunk r of every slice, in place) and
                 `all_gather_into_tensor` of the chunks when backward is done -- the two halves of a ring all-reduce as separate
                 calls, so that only the first half has to hide under backward.
  * "zero1":     the same reduce-scatter, but the all-gather moves PARAMETERS instead of gradients: `FusedAdamW` clips and
                 updates only this rank's chunk of every slice (1/N of the 30 B/param optimizer pass: 1.15 ms -> 0.14 ms at
                 N = 8, optimizer state effectively sharded), then the updated chunks are all-gathered -- the bf16 shadow the
                 GEMMs read for the layer buckets (2 B/param) and the f32 master for the last bucket (embeddings, norm weights,
                 visual embedding: read in f32 by the gather / norm kernels), slice by slice in the order the NEXT forward
                 needs them, each behind an event the engine's forward waits for (`vlt5_step.wait_events`).  Bytes on the links
                 per step and GPU, N = 8, bf16 buckets: reduce-scatter 7/8 x 451 MB + all-gather 7/8 x (401 + 99) MB = 0.83 GB,
                 against 2 x 7/8 x 451 MB = 0.79 GB for the all-reduce of the gradients alone and 1.58 GB for an f32 all-reduce.
                 The f32 master of the layer buckets follows behind the shadows on the same stream (`gather_master=True`, the
                 default: +4 B/param x 7/8 on the links per step, under the next forward, nothing waits for it but the next
                 update), so `state_dict()` / `flat_params()` / a rank-0-only checkpoint (the reference saves on rank 0 only,
                 vqacl.py:413-414, trainer_base.py:246-249) stay LOCAL calls: the Trainer is driven unchanged.
                 `gather_master=False` keeps the master of the layer buckets current only on the owning rank; then
                 `consolidate()` -- a collective every rank must call -- has to run before parameters are read, and
                 `state_dict()` raises until it has.  Needs `FusedAdamW`; with any other optimizer the wrapper completes the
                 gradients like "rs_ag".
  * "auto":      "zero1" when the world size divides 8 (every slice length is a multiple of 64 elements, so the chunks are
                 multiples of 8 elements = 16 bytes of bf16, what the chunk kernels read), "allreduce" otherwise.

`grad_dtype=torch.bfloat16` (default on the GPU) halves the bytes on the links: each merged slice travels as bf16 (the engine's
weight-gradient GEMMs write the bf16 staging copy themselves, `vlt5_step.grads_bf16`; only the last bucket is cast) and the
gradient-norm / AdamW kernels read the reduced bf16 values times 1/world straight from the staging buffer (`vlt5_sqnorm_g16`,
`vlt5_adamw_step_g16`).  `.grad` then holds the rank-local f32 gradients; `materialize_grads()` writes the averaged ones there.

Also all-reduces the prototype sufficient statistics (class sums and counts) so that every rank holds the
prototypes a single process would compute on the concatenated batch.
"""
import torch
import torch.distributed as dist

ALGOS = ("auto", "allreduce", "rs_ag", "zero1")


def pick_algo(world, algo="auto"):
    """The gradient exchange a wrapper over `world` ranks uses.  Slices are multiples of 64 elements, so chunks of a slice stay
    multiples of 8 elements (16 bytes of bf16, what the chunk kernels read) exactly when the world size divides 8: "auto" shards
    (zero1) for those and falls back to "allreduce" otherwise; asking for rs_ag / zero1 explicitly with any other world size raises."""
    if algo not in ALGOS:
        raise ValueError(f"algo must be one of {ALGOS}")
    world = int(world)
    if world < 1:
        raise ValueError("world size must be >= 1")
    divides = 8 % world == 0
    if algo == "auto":
        return "zero1" if (world > 1 and divides) else "allreduce"
    if algo != "allreduce" and not divides:
        raise ValueError("rs_ag / zero1 need a world size that divides 8 (chunks of a slice must stay 16-byte aligned in bf16)")
    return algo


def merge_buckets(bucket_start, bucket_end, lo, hi, bucket_bytes):
    """Buckets [lo, hi) merged into slices of >= bucket_bytes (f32 size): [(a, b, first_bucket, last_bucket)] in flat elements.  The
    remainder of a range stays a slice of its own on purpose: buckets are laid out in backward-completion order, so the remainder of the
    LAST released range is the bottom encoder layer -- the one reduce-scatter that cannot hide under backward and, under zero1, the first
    layer slice the next forward waits for: 14 MB there instead of 85 (tools/link_budget.py prints the plan)."""
    out, start = [], lo
    for b in range(lo, hi):
        size = (bucket_end[b] - bucket_start[start]) * 4
        if size >= bucket_bytes or b == hi - 1:
            out.append((bucket_start[start], bucket_end[b], start, b))
            start = b + 1
    return out


class DataParallelVLT5:
    def __init__(self, model, process_group=None, bucket_mb=128, average=True, grad_dtype=None, algo="auto", small_group=True,
                 gather_master=True):
        self.module = model
        if grad_dtype is None:
            grad_dtype = torch.bfloat16 if model._flat.is_cuda else torch.float32
        if grad_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("grad_dtype must be torch.float32 or torch.bfloat16")
        self.grad_dtype = grad_dtype
        self._g16 = None
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.algo = pick_algo(self.world, algo)
        self.gather_master = bool(gather_master)
        self.master_ready = None            # event behind the f32-master all-gather of the last sharded step (gather_master)
        self.average = average
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        model.dp = self
        # the latency-critical small collectives (prototype statistics in the middle of the forward, the squared gradient norm of the
        # sharded step) get a process group -- with RCCL: a communicator and stream -- of their own, so that they do not queue behind
        # hundreds of MB of reduce-scatter / all-gather traffic issued earlier on the bulk group
        self.ctrl_group = process_group
        if self.world > 1 and small_group:
            ranks = dist.get_process_group_ranks(process_group) if process_group is not None else None
            self.ctrl_group = dist.new_group(ranks=ranks)
        model.proto.dist_enabled = True
        model.proto.dist_group = self.ctrl_group
        # bucket b covers flat elements [start_b, end_b)
        ends = {}
        for name, (off, n, bucket, decay, used) in model._pinfo.items():
            if used:       # (rounded up to the 64-element alignment of the layout: the gap holds zeros and belongs to the bucket)
                ends[bucket] = max(ends.get(bucket, 0), (off + n + 63) // 64 * 64)
        self.bucket_end = [ends[b] for b in sorted(ends)]
        self.bucket_start = [0] + self.bucket_end[:-1]
        # The stream the waits for bucket events and the collectives are issued on: created on first use, its priority chosen per
        # configuration (the `comm_stream` property below)
        self._comm = None
        self._comm_prio = None
        self._events = None
        self.defer_cast_back = False        # set by FusedAdamW: the optimizer reads the reduced bf16 buckets itself
        self.g16_valid = False              # the staging buffer holds this backward's reduced gradients, not yet cast back
        self.sharded_optimizer = False      # set by FusedAdamW under algo == "zero1"
        self.shards_valid = False           # this backward left only this rank's chunks reduced (zero1, between backward and step)
        self.params_sharded = False         # the f32 master of the layer buckets is only current on the owning rank (zero1)
        self._slices_done = []              # slices reduce-scattered by the current backward, in issue order
        self._param_slices = []
        # the order the engine completes gradient buckets in (and the call that signals each): frozen here, equal on every rank
        self._frozen_plan = tuple(tuple(t) for t in model.grad_release_plan())
        if self.world > 1:
            # (a tensor collective on the model's device, like the weight broadcast below -- nothing the RCCL path does not do anyway)
            mine = torch.full((16,), -1, dtype=torch.int64, device=model._flat.device)
            flat_plan = [v for t in self._frozen_plan for v in t]
            mine[:len(flat_plan)] = torch.tensor(flat_plan, dtype=torch.int64)
            seen = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(seen, mine, group=process_group)
            if any(not torch.equal(t, mine) for t in seen):
                from ._lib import Vlt5Error
                raise Vlt5Error("ranks disagree on the gradient release plan (VLT5_* tuning variables differ between ranks?): "
                                + str([t.tolist() for t in seen]))
        # identical initial weights on every rank (what DDP's constructor would do)
        dist.broadcast(model._flat, src=0, group=process_group)
        model._bf16_version = -1

    def __getattr__(self, name):            # .train_step, .train(), .eval(), .state_dict() ... go to the model
        return getattr(self.__dict__["module"], name)

    # ---- the communication stream ----------------------------------------------------------------------
    def mirror_enabled(self):
        """bf16 buckets written by the engine itself (weight-gradient GEMM epilogues + vlt5_mirror_rows_bf16: vlt5_step.grads_bf16), so
        that no cast pass runs on the communication stream.  VQACL_DP_MIRROR=0 turns it off (the cast-per-slice path of round 1)."""
        import os
        return self.grad_dtype is torch.bfloat16 and self.module._flat.is_cuda and os.environ.get("VQACL_DP_MIRROR", "1") != "0"

    def comm_carries_kernels(self):
        """Does this configuration enqueue KERNELS on the communication stream (besides event waits and collectives)?  f32 buckets:
        the 1/world scaling (`div_`); bf16 buckets without the engine's mirror: `vlt5_cast_bf16` per slice; bf16 buckets without
        FusedAdamW reading the staging buffer (`defer_cast_back`): `vlt5_cast_f32` behind every all-reduce / all-gather."""
        return not (self.mirror_enabled() and self.defer_cast_back)

    def comm_priority(self):
        """0 (normal) when the stream carries event waits only: a HIGH-priority stream sitting on a pending event wait costs the compute
        queue +0.23 ms per step at world size 1 on this runtime, and a persistent 3x slowdown when the GPU was touched before
        init_process_group() (profiles/r05_w_comm_stream_priority.txt).  -1 (high) when it carries kernels: at normal priority the casts
        of a bucket released mid-backward only ran once backward had finished (round 1, tools/dp_overlap_probe.py) -- the overlap the
        buckets exist for.  VQACL_COMM_PRIORITY overrides.  (The collectives themselves run on the process group's internal stream:
        create the group with `ProcessGroupNCCL.Options(is_high_priority_stream=True)`, as bench.py does -- INTEGRATION.md.)"""
        import os
        forced = os.environ.get("VQACL_COMM_PRIORITY")
        if forced is not None:
            return int(forced)
        return -1 if self.comm_carries_kernels() else 0

    @property
    def comm_stream(self):
        if not self.module._flat.is_cuda:
            return None
        want = self.comm_priority()
        if self._comm is None or self._comm_prio != want:
            new = torch.cuda.Stream(device=self.module._flat.device, priority=want)     # (first use may be on an autograd thread)
            if self._comm is not None:      # the configuration changed (an optimizer was attached, a switch flipped): keep the order
                new.wait_stream(self._comm)
            self._comm, self._comm_prio = new, want
        return self._comm

    def describe(self):
        return {"algo": self.algo, "grad_dtype": str(self.grad_dtype).replace("torch.", ""), "world": self.world,
                "bucket_mb": self.bucket_bytes >> 20, "sharded_optimizer": bool(self.sharded_optimizer),
                "gather_master": bool(self.gather_master)}

    @property
    def release_ranges(self):
        """The bucket ranges the engine releases gradients in, in order (VLT5.grad_release_plan): stacked cross k/v, decoder layers,
        upper half of the encoder, the last bucket (embeddings / norms), lower half of the encoder -- the slice plan, and with it the
        chunk every rank owns under zero1, only depends on these.  FROZEN when the wrapper is built (`_frozen_plan`): the owned chunks
        and the sharded Adam moments follow from it, so it must not move with a later edit of `model.tuning` / the side-stream switch."""
        return tuple((lo, hi) for _, lo, hi in self._frozen_plan)

    def check_release_plan(self, plan):
        """Called by every backward with the plan the engine reports NOW: a plan that moved since the wrapper was built would silently
        change chunk ownership (stale Adam moments) and the call a bucket's event is recorded by (a stale event) -- refuse."""
        if tuple(tuple(t) for t in plan) != self._frozen_plan:
            from ._lib import Vlt5Error
            raise Vlt5Error(f"the gradient release plan changed after DataParallelVLT5 was built ({self._frozen_plan} -> {tuple(map(tuple, plan))}): "
                            "model.tuning (wgrad_shadow, enc_cut) and the side-stream switch must be set BEFORE the wrapper is created; "
                            "build a new wrapper (and optimizer) for a different plan")

    def slice_plan(self):
        """Every merged slice a backward reduce-scatters, [(a, b)] in flat elements, in issue order."""
        return [(a, b) for lo, hi in self.release_ranges for a, b, _, _ in self.slices_of(lo, hi)]

    # ---- slice plan -----------------------------------------------------------------------------------
    def slices_of(self, lo, hi):
        """Buckets [lo, hi) merged into slices of >= bucket_bytes: [(a, b, first_bucket, last_bucket)] in flat elements."""
        return merge_buckets(self.bucket_start, self.bucket_end, lo, hi, self.bucket_bytes)

    def chunk(self, a, b, rank=None):
        """Rank `rank`'s chunk of slice [a, b): equal parts (slice lengths are multiples of 64 elements)."""
        c = (b - a) // self.world
        r = self.rank if rank is None else rank
        return a + r * c, a + (r + 1) * c

    # ---- called by VLT5._engine_backward -------------------------------------------------------------
    def make_events(self, n):
        if self._events is None or len(self._events) != n:
            self._events = [torch.cuda.Event() for _ in range(n)]
            for e in self._events:          # force creation of the underlying hipEvent_t
                e.record()
        self._slices_done = []
        self.shards_valid = False
        return self._events

    @property
    def grad_scale(self):
        return 1.0 / self.world if self.average else 1.0

    def materialize_grads(self, model=None):
        """Deferred mode: write the averaged gradients into the f32 gradient buffer (what `.grad` views) on the current stream.
        (zero1: between backward and the optimizer step only this rank's chunks are reduced; the chunks are all-gathered first --
        a collective, every rank must call it.  The rank's own chunks stay valid, so the sharded optimizer step still runs.)"""
        if self.shards_valid:
            self._allgather_grads((model or self.module)._flat_grad, keep=self.sharded_optimizer)
            if self.comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        if not self.g16_valid:
            return
        from ._lib import check, lib, ptr, stream_ptr
        flat = (model or self.module)._flat_grad
        end = self.bucket_end[-1]
        check(lib().vlt5_cast_f32(ptr(self._g16), ptr(flat), end, self.grad_scale, stream_ptr()), "vlt5_cast_f32")
        self.g16_valid = False

    def staging(self, flat):
        """The bf16 mirror of the flat gradient buffer (same element offsets).  Zero-initialised once: alignment gaps are never
        written and must reduce to zero."""
        if self._g16 is None or self._g16.numel() < flat.numel():
            self._g16 = torch.zeros(flat.numel(), device=flat.device, dtype=torch.bfloat16)
        return self._g16

    def _wire(self, flat):
        """The buffer that travels: the bf16 staging mirror (GPU, bf16 buckets) or the f32 gradient buffer itself."""
        return self.staging(flat) if (self.grad_dtype is torch.bfloat16 and flat.is_cuda) else flat

    def _reduce_slice(self, flat, a, b, defer=False, mirrored=False):
        """One merged slice on the current (comm) stream: all-reduce, or reduce-scatter leaving chunk `rank` reduced in place."""
        t = flat[a:b]
        bf16 = self.grad_dtype is torch.bfloat16 and flat.is_cuda
        if bf16:
            from ._lib import check, lib, ptr, stream_ptr
            h = self.staging(flat)[a:b]
            if not mirrored:        # (mirrored: the engine's weight-gradient GEMMs wrote the bf16 copy themselves: vlt5_step.grads_bf16)
                check(lib().vlt5_cast_bf16(ptr(t), ptr(h), b - a, stream_ptr()), "vlt5_cast_bf16")
            wire = h
        else:
            wire = t
        if self.algo == "allreduce":
            dist.all_reduce(wire, group=self.group)
        else:
            ca, cb = self.chunk(a, b)
            dist.reduce_scatter_tensor(wire[ca - a:cb - a], wire, group=self.group)
            if not bf16 and self.average:
                wire[ca - a:cb - a].div_(self.world)      # (bf16 chunks are scaled by the kernels that read them: grad_scale)
            self._slices_done.append((a, b))
            return
        if bf16:
            if not defer:
                check(lib().vlt5_cast_f32(ptr(h), ptr(t), b - a, self.grad_scale, stream_ptr()), "vlt5_cast_f32")
        elif self.average:
            t.div_(self.world)

    def _allgather_grads(self, flat, keep=False):
        """Second half of rs_ag (and the fall-back of zero1 without a sharded optimizer): every rank receives every chunk.
        keep: the reduce-scattered state (slice list, shards_valid) survives -- the sharded optimizer step that follows reads
        this rank's chunks, which the all-gather leaves as they are."""
        wire = self._wire(flat)
        bf16 = wire is not flat
        ctx = torch.cuda.stream(self.comm_stream) if self.comm_stream is not None else _null()
        with ctx:
            for a, b in self._slices_done:
                ca, cb = self.chunk(a, b)
                dist.all_gather_into_tensor(wire[a:b], wire[ca:cb], group=self.group)
                if bf16:
                    if not (self.defer_cast_back and self.grad_dtype is torch.bfloat16):
                        from ._lib import check, lib, ptr, stream_ptr
                        check(lib().vlt5_cast_f32(ptr(wire[a:b]), ptr(flat[a:b]), b - a, self.grad_scale, stream_ptr()), "vlt5_cast_f32")
        if not keep:
            self._slices_done = []
            self.shards_valid = False

    def reduce_range(self, model, events, lo, hi, final=False, mirrored=False):
        """Issue the collectives of buckets [lo, hi) on the comm stream: consecutive buckets are merged up to `bucket_bytes`, each
        merged slice goes after the event of its last bucket.  `final`: this call completes the gradient buffer."""
        flat = model._flat_grad
        defer = self.defer_cast_back and self.grad_dtype is torch.bfloat16
        with torch.cuda.stream(self.comm_stream):
            for a, b, first, last in self.slices_of(lo, hi):
                self.comm_stream.wait_event(events[last])
                self._reduce_slice(flat, a, b, defer=defer, mirrored=mirrored)
        if final:
            self.g16_valid = defer

    def finish(self):
        """End of backward.  allreduce: nothing left; rs_ag (or zero1 without its optimizer): all-gather the reduced chunks;
        zero1: the chunks stay as they are for the sharded optimizer step."""
        if self.algo != "allreduce":
            if self.algo == "zero1" and self.sharded_optimizer:
                self.shards_valid = True
            else:
                self._allgather_grads(self.module._flat_grad)
        torch.cuda.current_stream().wait_stream(self.comm_stream)

    def reduce_flat(self, flat):
        """Non-overlapped path (gradient accumulation into a temporary buffer, or CPU/gloo tests): one all-reduce of everything."""
        end = self.bucket_end[-1]
        self.g16_valid = False
        self.shards_valid = False
        algo, self.algo = self.algo, "allreduce"
        try:
            self._reduce_slice(flat, 0, end)
        finally:
            self.algo = algo

    # ---- zero1: parameters after the sharded update ---------------------------------------------------
    def owned_ranges(self):
        """This rank's chunk of every slice reduce-scattered by the last backward: [(a, b)] in flat elements."""
        return [self.chunk(a, b) for a, b in self._slices_done]

    def allgather_updated(self, model, slices, main_event):
        """After the sharded optimizer updated chunk `rank` of `slices` (list of (a, b)): all-gather the chunks of the bf16 shadow
        (layer buckets) or of the f32 master + shadow (last bucket) on the comm stream, behind `main_event` (recorded on the
        optimizer's stream after the update of these slices).  Returns nothing; the caller records the per-bucket events."""
        last_a = self.bucket_start[-1]
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(main_event)
            for a, b in slices:
                ca, cb = self.chunk(a, b)
                if a >= last_a:      # embeddings / norm weights / visual embedding: the kernels read the f32 master
                    dist.all_gather_into_tensor(model._flat[a:b], model._flat[ca:cb], group=self.group)
                dist.all_gather_into_tensor(model._flat_bf16[a:b], model._flat_bf16[ca:cb], group=self.group)

    def gather_master_async(self, model, slices):
        """zero1 with gather_master: all-gather the f32 master chunks of the layer buckets on the comm stream, BEHIND the shadow
        all-gathers the next forward waits for.  Every rank issues it as part of its optimizer step, so reading parameters
        afterwards (state_dict(), a rank-0-only checkpoint) is a local wait for `master_ready`, never a collective."""
        last_a = self.bucket_start[-1]
        with torch.cuda.stream(self.comm_stream):
            for a, b in slices:
                if a >= last_a:
                    continue            # embeddings / norms: gathered in f32 together with their shadow
                ca, cb = self.chunk(a, b)
                dist.all_gather_into_tensor(model._flat[a:b], model._flat[ca:cb], group=self.group)
            if self.master_ready is None:
                self.master_ready = torch.cuda.Event()
            self.master_ready.record(self.comm_stream)

    def consolidate(self, model=None):
        """zero1 with gather_master=False: make the f32 master complete on this rank (all-gather of every rank's chunks).
        COLLECTIVE -- every rank must call it, at the same point of the program; `state_dict()` refuses to run before."""
        if not self.params_sharded:
            return
        model = model or self.module
        model.sync_optimizer()
        in_sync = model._bf16_version == model._flat._version
        last_a = self.bucket_start[-1]
        for a, b in self._param_slices:
            if a >= last_a:
                continue            # already gathered in f32 every step
            ca, cb = self.chunk(a, b)
            dist.all_gather_into_tensor(model._flat[a:b], model._flat[ca:cb], group=self.group)
        self.params_sharded = False
        if in_sync:                 # the shadow already equals bf16(master) everywhere; a pending refresh (load_state_dict,
            model._bf16_version = model._flat._version     # p.data.copy_ since the last step) stays pending

    materialize_params = consolidate


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
