"""Host side of the SS/SI prototype head (reference: VL-T5/src/modeling_t5_our.py:434-511, 583-615).

The arithmetic runs in the HIP kernels of csrc/proto.hip; this class only keeps the per-task state machine
of `VLT5.update_prototype` (which task has been seen, which task owns a memory tensor) -- plain Python control
flow on the integer `current_task_id`, exactly like the reference, so no device->host synchronisation is needed.
"""
import torch

from . import _lib as L
from ._lib import check, lib, ptr, stream_ptr
from . import ops


class PrototypeHead:
    def __init__(self, n_ques: int, n_cate: int, d_model: int, device):
        self.CQ, self.CV, self.d = n_ques, n_cate, d_model
        self.device = device
        z = dict(device=device, dtype=torch.float32)
        self.Q_prototype = torch.zeros(n_ques, d_model, **z)
        self.V_prototype = torch.zeros(n_cate, d_model, **z)
        self.Q_prototype_num = torch.zeros(n_ques, **z)
        self.V_prototype_num = torch.zeros(n_cate, **z)
        self.seen_tasks = set()            # keys of the reference's Q_task_cur_proto
        self.Q_task_mem_proto = {}         # task -> [CQ,d] memory tensor (reference name)
        self.dist_enabled = False          # set by DataParallelVLT5: statistics are summed over ranks ...
        self.dist_group = None             # ... of this process group (None = the default group)

    def reset(self):
        self.seen_tasks.clear()
        self.Q_task_mem_proto.clear()

    # ---- training: calculate_current_prototype x2 + update_prototype ---------------------------------
    def update(self, poolQ, poolV, ques_labels, cate_labels, task: int, alpha: float, beta: float):
        curQ, numQ = ops.proto_class_mean(poolQ, ques_labels)
        curV, numV = ops.proto_class_mean(poolV, cate_labels)
        if self.dist_enabled:
            (curQ, numQ), (curV, numV) = self._allreduce_stats((curQ, numQ), (curV, numV))
        first = task not in self.seen_tasks
        qmem, qinit = None, 0
        if not first and task != 0:
            if task in self.Q_task_mem_proto:
                qmem, qinit = self.Q_task_mem_proto[task], 1
            else:
                qmem = torch.empty_like(self.Q_prototype)
                self.Q_task_mem_proto[task] = qmem
        check(lib().vlt5_proto_update(ptr(curQ), ptr(curV), ptr(numQ), ptr(numV), ptr(self.Q_prototype), ptr(self.V_prototype),
                                      ptr(self.Q_prototype_num), ptr(self.V_prototype_num), ptr(qmem), qinit, int(first),
                                      int(task), float(alpha), float(beta), self.CQ, self.CV, self.d, stream_ptr()),
              "vlt5_proto_update")
        self.seen_tasks.add(task)
        return curQ, curV

    def _allreduce_stats(self, *stats):
        """Data parallel: class means over the GLOBAL batch = all-reduced sums / all-reduced counts, so every rank
        holds the prototypes a single process would compute on the concatenated batch.  The Q and V statistics
        travel in ONE collective (277 KB): a small all-reduce is pure latency.  Packing (mean x count | count) and unpacking are one
        launch each (vlt5_proto_stats_pack); on the CPU (gloo tests of the host logic) the same arithmetic in torch."""
        import torch.distributed as dist
        if len(stats) == 2 and stats[0][0].is_cuda:
            (curQ, numQ), (curV, numV) = stats
            n = (self.CQ + self.CV) * (self.d + 1)
            if getattr(self, "_packed", None) is None or self._packed.numel() != n:
                self._packed = torch.empty(n, device=curQ.device, dtype=torch.float32)
            args = (ptr(curQ), ptr(numQ), ptr(curV), ptr(numV), ptr(self._packed), self.CQ, self.CV, self.d)
            check(lib().vlt5_proto_stats_pack(*args, 0, stream_ptr()), "vlt5_proto_stats_pack")
            dist.all_reduce(self._packed, group=self.dist_group)
            check(lib().vlt5_proto_stats_pack(*args, 1, stream_ptr()), "vlt5_proto_stats_pack")
            return (curQ, numQ), (curV, numV)
        parts = []
        for proto, cnt in stats:
            parts += [(proto * cnt.clamp(min=1).unsqueeze(1)).reshape(-1), cnt]
        packed = torch.cat(parts)
        dist.all_reduce(packed, group=self.dist_group)
        out, off = [], 0
        for proto, cnt in stats:
            n = proto.numel()
            sums = packed[off:off + n].view_as(proto)
            c = packed[off + n:off + n + cnt.numel()].clone()
            off += n + cnt.numel()
            out.append(((sums / c.clamp(min=1).unsqueeze(1)).contiguous(), c))
        return out

    # ---- retrieval: cosine_similarity_multi + gather, written straight into the decoder's memory rows --
    def retrieve(self, poolQ, poolV, enc_f32, enc_bf16, S: int):
        """enc_f32 / enc_bf16: [B, S+2, d] buffers whose rows S and S+1 receive the retrieved Q / V prototypes."""
        Sx, d = S + 2, self.d
        idxQ = ops.proto_retrieve(self.Q_prototype, poolQ, enc_f32[:, S], Sx * d, enc_bf16[:, S], Sx * d)
        idxV = ops.proto_retrieve(self.V_prototype, poolV, enc_f32[:, S + 1], Sx * d, enc_bf16[:, S + 1], Sx * d)
        return idxQ, idxV

    # ---- the whole head of one forward: pooling, (update), retrieval of both heads -- three launches (vlt5_proto_head_fwd) ----
    def forward(self, enc_f32, enc_bf16, S: int, split: int, ques_labels=None, cate_labels=None, task: int = 0, alpha: float = 0.5,
                beta: float = 0.3, update: bool = False):
        """enc_f32 / enc_bf16: [B, S+2, d] decoder-memory buffers (rows 0..S-1 = encoder output, rows S, S+1 receive the retrieved
        prototypes).  Returns (poolQ, poolV, idxQ, idxV).  Single process only: under data parallelism the class statistics are
        all-reduced between the class means and the state update (`update` + `retrieve`)."""
        import ctypes as C
        assert not (update and self.dist_enabled)
        B, Sx, d = enc_f32.shape
        dev = enc_f32.device
        h = L.ProtoHeadDesc()
        poolQ = torch.empty(B, d, device=dev, dtype=torch.float32)
        poolV = torch.empty(B, d, device=dev, dtype=torch.float32)
        idxQ = torch.empty(B, device=dev, dtype=torch.int64)
        idxV = torch.empty(B, device=dev, dtype=torch.int64)
        if getattr(self, "_scratch", None) is None:
            self._scratch = torch.empty((self.CQ + self.CV) * d, device=dev, dtype=torch.float32)
        h.hidden, h.hidden_sb, h.B, h.S, h.d, h.split = ptr(enc_f32), enc_f32.stride(0), B, S, d, split
        h.poolQ, h.poolV = ptr(poolQ), ptr(poolV)
        h.Qproto, h.Vproto, h.Qnum, h.Vnum = ptr(self.Q_prototype), ptr(self.V_prototype), ptr(self.Q_prototype_num), ptr(self.V_prototype_num)
        h.CQ, h.CV, h.alpha, h.beta = self.CQ, self.CV, float(alpha), float(beta)
        h.idxQ, h.idxV = ptr(idxQ), ptr(idxV)
        h.out_f32, h.out_sb = ptr(enc_f32[:, S]), Sx * d
        h.out_bf16, h.out_sb_bf16 = ptr(enc_bf16[:, S]), Sx * d
        h.scratch = ptr(self._scratch)
        keep = (poolQ, poolV, idxQ, idxV)
        if update:
            ql, cl = ques_labels.contiguous(), cate_labels.contiguous()
            first = task not in self.seen_tasks
            qmem, qinit = None, 0
            if not first and task != 0:
                if task in self.Q_task_mem_proto:
                    qmem, qinit = self.Q_task_mem_proto[task], 1
                else:
                    qmem = torch.empty_like(self.Q_prototype)
                    self.Q_task_mem_proto[task] = qmem
            h.onehotQ, h.onehotV, h.qmem, h.qmem_initialised = ptr(ql), ptr(cl), ptr(qmem), qinit
            h.first, h.task, h.update = int(first), int(task), 1
            keep = keep + (ql, cl, qmem)
        check(lib().vlt5_proto_head_fwd(C.byref(h), stream_ptr()), "vlt5_proto_head_fwd")
        if update:
            self.seen_tasks.add(task)
        del keep
        return poolQ, poolV, idxQ, idxV

    def forward_dist(self, enc_f32, enc_bf16, S: int, split: int, ques_labels, cate_labels, task: int, alpha: float, beta: float):
        """The training head under data parallelism in two halves around ONE all-reduce (vlt5_proto_head_desc.phase, round 5): pooling +
        the local batch's class sums and counts -> `packed`; all-reduce; class means of the GLOBAL batch, the state update, the normalised
        copies and the retrieval of both heads -- four launches instead of ten, and every rank holds the prototypes a single process
        would compute on the concatenated batch.  Returns (poolQ, poolV, idxQ, idxV)."""
        import ctypes as C
        import torch.distributed as dist
        B, Sx, d = enc_f32.shape
        dev = enc_f32.device
        h = L.ProtoHeadDesc()
        poolQ = torch.empty(B, d, device=dev, dtype=torch.float32)
        poolV = torch.empty(B, d, device=dev, dtype=torch.float32)
        idxQ = torch.empty(B, device=dev, dtype=torch.int64)
        idxV = torch.empty(B, device=dev, dtype=torch.int64)
        if getattr(self, "_scratch", None) is None:
            self._scratch = torch.empty((self.CQ + self.CV) * d, device=dev, dtype=torch.float32)
        n = (self.CQ + self.CV) * (d + 1)
        if getattr(self, "_packed", None) is None or self._packed.numel() != n:
            self._packed = torch.empty(n, device=dev, dtype=torch.float32)
        ql, cl = ques_labels.contiguous(), cate_labels.contiguous()
        first = task not in self.seen_tasks
        qmem, qinit = None, 0
        if not first and task != 0:
            if task in self.Q_task_mem_proto:
                qmem, qinit = self.Q_task_mem_proto[task], 1
            else:
                qmem = torch.empty_like(self.Q_prototype)
                self.Q_task_mem_proto[task] = qmem
        h.hidden, h.hidden_sb, h.B, h.S, h.d, h.split = ptr(enc_f32), enc_f32.stride(0), B, S, d, split
        h.poolQ, h.poolV = ptr(poolQ), ptr(poolV)
        h.Qproto, h.Vproto, h.Qnum, h.Vnum = ptr(self.Q_prototype), ptr(self.V_prototype), ptr(self.Q_prototype_num), ptr(self.V_prototype_num)
        h.CQ, h.CV, h.alpha, h.beta = self.CQ, self.CV, float(alpha), float(beta)
        h.idxQ, h.idxV = ptr(idxQ), ptr(idxV)
        h.out_f32, h.out_sb = ptr(enc_f32[:, S]), Sx * d
        h.out_bf16, h.out_sb_bf16 = ptr(enc_bf16[:, S]), Sx * d
        h.scratch, h.packed = ptr(self._scratch), ptr(self._packed)
        h.onehotQ, h.onehotV, h.qmem, h.qmem_initialised = ptr(ql), ptr(cl), ptr(qmem), qinit
        h.first, h.task, h.update = int(first), int(task), 1
        h.phase = 1
        check(lib().vlt5_proto_head_fwd(C.byref(h), stream_ptr()), "vlt5_proto_head_fwd (phase 1)")
        dist.all_reduce(self._packed, group=self.dist_group)
        h.phase = 2
        check(lib().vlt5_proto_head_fwd(C.byref(h), stream_ptr()), "vlt5_proto_head_fwd (phase 2)")
        self.seen_tasks.add(task)
        del ql, cl, qmem
        return poolQ, poolV, idxQ, idxV

    def memory_loss(self, poolQ, poolV, ques_labels, cate_labels):
        return (ops.proto_memory_loss(poolQ, ques_labels, self.Q_prototype),
                ops.proto_memory_loss(poolV, cate_labels, self.V_prototype))
