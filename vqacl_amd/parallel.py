"""Data parallelism for the engine-backed VLT5: one process per GPU, RCCL all-reduce over xGMI.

The reference wraps the model in DDP but calls `model.module.train_step`, which bypasses DDP's reducer, so its
ranks never synchronise gradients.  This wrapper provides the synchronisation the north star asks
for: because the gradients live in ONE flat buffer laid out in backward-completion order, a bucket is just a
slice -- no flatten/unflatten copies.  The engine records a HIP event when each layer's gradients are complete;
the comm stream waits on the event and all-reduces that slice while the compute stream continues with the next
layer (overlap with backward).  xGMI is point-to-point, so few, large collectives are preferred: consecutive
layer buckets are merged up to `bucket_mb`.

`grad_dtype=torch.bfloat16` (default on the GPU) halves the bytes on the links: each merged bucket is cast to a bf16 staging
buffer, all-reduced in bf16 and cast back (scaled by 1/world) into the fp32 gradient buffer on the comm stream, so clipping,
the optimizer and `.grad` see fp32 as before.  At 8 GPUs the fp32 reduction of 0.9 GB is ~5 ms of ring time per step against
~6.5 ms of backward to hide it in; bf16 needs half.  (The rounding -- one bf16 rounding per rank contribution -- is of the same
size as the bf16 GEMM noise already in the gradients.)

With the fused optimizer the way back is not a pass of its own: `FusedAdamW` attached to a wrapped model sets `defer_cast_back`,
the reduced buckets stay in the bf16 staging buffer and the gradient-norm and AdamW kernels read them from there
(`vlt5_sqnorm_g16` / `vlt5_adamw_step_g16`, gradient = bf16 * 1/world: the very values the cast back would have written), which
takes 1.35 GB of HBM traffic and one kernel per bucket out of the step's tail.  `.grad` then still holds the rank-local f32
gradients; `materialize_grads()` writes the averaged ones there on demand.

Also all-reduces the prototype sufficient statistics (class sums and counts) so that every rank holds the
prototypes a single process would compute on the concatenated batch.
"""
import torch
import torch.distributed as dist


class DataParallelVLT5:
    def __init__(self, model, process_group=None, bucket_mb=128, average=True, grad_dtype=None):
        self.module = model
        if grad_dtype is None:
            grad_dtype = torch.bfloat16 if model._flat.is_cuda else torch.float32
        if grad_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("grad_dtype must be torch.float32 or torch.bfloat16")
        self.grad_dtype = grad_dtype
        self._g16 = None
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.average = average
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        model.dp = self
        model.proto.dist_enabled = True
        model.proto.dist_group = process_group
        # bucket b covers flat elements [start_b, end_b)
        ends = {}
        for name, (off, n, bucket, decay, used) in model._pinfo.items():
            if used:
                ends[bucket] = max(ends.get(bucket, 0), off + n)
        self.bucket_end = [ends[b] for b in sorted(ends)]
        self.bucket_start = [0] + self.bucket_end[:-1]
        self.comm_stream = torch.cuda.Stream(priority=-1) if model._flat.is_cuda else None   # high priority: short casts + collectives must not queue behind the backward GEMMs
        self._events = None
        self._next = 0
        self._pending_from = 0
        self.defer_cast_back = False        # set by FusedAdamW: the optimizer reads the reduced bf16 buckets itself
        self.g16_valid = False              # the staging buffer holds this backward's reduced gradients, not yet cast back
        # identical initial weights on every rank (what DDP's constructor would do)
        dist.broadcast(model._flat, src=0, group=process_group)
        model._bf16_version = -1

    def __getattr__(self, name):            # .train_step, .train(), .eval(), .state_dict() ... go to the model
        return getattr(self.__dict__["module"], name)

    # ---- called by VLT5._engine_backward -------------------------------------------------------------
    def make_events(self, n):
        if self._events is None or len(self._events) != n:
            self._events = [torch.cuda.Event() for _ in range(n)]
            for e in self._events:          # force creation of the underlying hipEvent_t
                e.record()
        self._next = 0
        self._pending_from = 0
        return self._events

    @property
    def grad_scale(self):
        return 1.0 / self.world if self.average else 1.0

    def materialize_grads(self, model=None):
        """Deferred mode: write the averaged gradients into the f32 gradient buffer (what `.grad` views) on the current stream."""
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

    def _allreduce_slice(self, flat, a, b, defer=False, mirrored=False):
        t = flat[a:b]
        if self.grad_dtype is torch.bfloat16 and flat.is_cuda:
            from ._lib import check, lib, ptr, stream_ptr
            h = self.staging(flat)[a:b]
            if not mirrored:        # (mirrored: the engine's weight-gradient GEMMs wrote the bf16 copy themselves: vlt5_step.grads_bf16)
                check(lib().vlt5_cast_bf16(ptr(t), ptr(h), b - a, stream_ptr()), "vlt5_cast_bf16")
            dist.all_reduce(h, group=self.group)
            if not defer:
                check(lib().vlt5_cast_f32(ptr(h), ptr(t), b - a, self.grad_scale, stream_ptr()), "vlt5_cast_f32")
            return
        dist.all_reduce(t, group=self.group)
        if self.average:
            t.div_(self.world)

    def reduce_range(self, model, events, lo, hi, final=False, mirrored=False):
        """Issue the all-reduces of buckets [lo, hi) on the comm stream: consecutive buckets are merged up to `bucket_bytes`, each
        merged slice goes after the event of its last bucket.  `final`: this call completes the gradient buffer."""
        flat = model._flat_grad
        defer = self.defer_cast_back and self.grad_dtype is torch.bfloat16
        with torch.cuda.stream(self.comm_stream):
            start = lo
            for b in range(lo, hi):
                size = (self.bucket_end[b] - self.bucket_start[start]) * 4
                if size >= self.bucket_bytes or b == hi - 1:
                    self.comm_stream.wait_event(events[b])
                    self._allreduce_slice(flat, self.bucket_start[start], self.bucket_end[b], defer=defer, mirrored=mirrored)
                    start = b + 1
        if final:
            self.g16_valid = defer

    def finish(self):
        torch.cuda.current_stream().wait_stream(self.comm_stream)

    def reduce_flat(self, flat):
        """Non-overlapped path (gradient accumulation into a temporary buffer, or CPU/gloo tests)."""
        end = self.bucket_end[-1]
        self.g16_valid = False
        self._allreduce_slice(flat, 0, end)
