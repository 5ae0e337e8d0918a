"""VL-T5 model whose forward/backward run on the hand-written gfx950 kernels (libvlt5_hip.so).

Mirrors the surface of the reference's `VLT5` (VL-T5/src/modeling_t5_our.py:342-772): same module tree /
state_dict names, same `forward(...)` keyword arguments, same output fields, same prototype attributes
(`Q_prototype`, `V_prototype`), so `vqa_model.VLT5VQA.train_step` and the `Trainer` drive it unchanged.
PyTorch is plumbing only: it owns the device memory, hooks the engine into autograd with ONE node per step,
and runs whatever optimizer the Trainer built over `named_parameters()`.

Memory layout (sized for 288 GB HBM, one process per GPU):
  * all parameters are views into ONE flat fp32 buffer (master weights) ordered so that gradients complete
    front-to-back during backward; a flat bf16 shadow with identical offsets feeds the MFMA GEMMs; a flat fp32
    gradient buffer receives the grads (so data-parallel all-reduce works on a few large contiguous buckets);
  * activations live in one workspace arena planned by the C side (`vlt5_workspace_bytes`).
"""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from ._lib import Config as CConfig, Step as CStep, check, lib, ptr, stream_ptr
from .buckets import bucket_table
from .prototype import PrototypeHead
from . import ops


class VLT5Config:
    """Hyper-parameters read from the reference's config object (trainer_base.py:57-89) or given directly."""

    def __init__(self, d_model=768, d_kv=64, num_heads=12, d_ff=3072, num_layers=12, num_decoder_layers=None,
                 vocab_size=32200, relative_attention_num_buckets=32, layer_norm_epsilon=1e-6, dropout_rate=0.1,
                 feat_dim=2048, pos_dim=4, n_images=2, pad_token_id=0, decoder_start_token_id=0, n_ques=10, n_cate=80,
                 feed_forward_proj="relu", tie_word_embeddings=True, classifier=False, **unused):
        self.d_model, self.d_kv, self.num_heads, self.d_ff = d_model, d_kv, num_heads, d_ff
        self.num_layers = num_layers
        self.num_decoder_layers = num_decoder_layers if num_decoder_layers is not None else num_layers
        self.vocab_size = vocab_size
        self.relative_attention_num_buckets = relative_attention_num_buckets
        self.layer_norm_epsilon = layer_norm_epsilon
        self.dropout_rate = dropout_rate
        self.feat_dim, self.pos_dim, self.n_images = feat_dim, pos_dim, n_images
        self.pad_token_id, self.decoder_start_token_id = pad_token_id, decoder_start_token_id
        self.n_ques, self.n_cate = n_ques, n_cate
        self.feed_forward_proj = feed_forward_proj
        self.tie_word_embeddings = tie_word_embeddings
        self.classifier = classifier
        # HF T5Config: "relu" (t5-base / t5-large: every reference launch script) or "gated-gelu" (t5-v1.1: T5DenseGatedActDense)
        if feed_forward_proj not in ("relu", "gated-gelu"):
            raise NotImplementedError(f"feed_forward_proj={feed_forward_proj!r}: the engine implements 'relu' and 'gated-gelu'")
        self.is_gated_act = feed_forward_proj == "gated-gelu"
        if pos_dim != 4 or not tie_word_embeddings:
            raise NotImplementedError("pos_dim must be 4 and embeddings tied, as in every reference launch script")

    @classmethod
    def from_any(cls, cfg):
        if isinstance(cfg, cls):
            return cfg
        keys = ("d_model d_kv num_heads d_ff num_layers num_decoder_layers vocab_size relative_attention_num_buckets "
                "layer_norm_epsilon dropout_rate feat_dim pos_dim n_images pad_token_id decoder_start_token_id "
                "feed_forward_proj tie_word_embeddings classifier n_ques n_cate").split()
        kw = {k: getattr(cfg, k) for k in keys if getattr(cfg, k, None) is not None}
        return cls(**kw)

    def c_struct(self):
        c = CConfig()
        c.d_model, c.d_kv, c.num_heads, c.d_ff = self.d_model, self.d_kv, self.num_heads, self.d_ff
        c.num_layers, c.num_decoder_layers, c.vocab = self.num_layers, self.num_decoder_layers, self.vocab_size
        c.rel_buckets, c.feat_dim, c.n_images = self.relative_attention_num_buckets, self.feat_dim, self.n_images
        c.pad_id, c.dec_start_id = self.pad_token_id, self.decoder_start_token_id
        c.n_ques, c.n_cate = self.n_ques, self.n_cate
        c.eps, c.dropout = self.layer_norm_epsilon, self.dropout_rate
        c.gated_act = int(self.is_gated_act)
        return c


def param_layout(cfg: VLT5Config):
    """[(name, offset, shape, bucket, decay, used)] of the flat parameter buffer, from the C side."""
    c = cfg.c_struct()
    n = lib().vlt5_layout_count(C.byref(c))
    out = []
    buf = C.create_string_buffer(160)
    off, rows, cols, bucket, decay, used = L.c_ll(), L.c_i(), L.c_i(), L.c_i(), L.c_i(), L.c_i()
    for i in range(n):
        check(lib().vlt5_layout_get(C.byref(c), i, buf, 160, C.byref(off), C.byref(rows), C.byref(cols), C.byref(bucket),
                                    C.byref(decay), C.byref(used)), "vlt5_layout_get")
        shape = (rows.value, cols.value) if cols.value > 0 else (rows.value,)
        out.append((buf.value.decode(), off.value, shape, bucket.value, bool(decay.value), bool(used.value)))
    return out, lib().vlt5_layout_total(C.byref(c)), lib().vlt5_layout_buckets(C.byref(c))


class _Holder(nn.Module):
    """Parameter container node of the module tree (names only; the arithmetic runs in the engine)."""

    def forward(self, *a, **k):
        raise RuntimeError("sub-modules of the engine-backed VLT5 hold parameters only; call the model itself")


class _Linear(nn.Linear):
    def forward(self, *a, **k):
        raise RuntimeError("sub-modules of the engine-backed VLT5 hold parameters only; call the model itself")


class _Embedding(nn.Embedding):
    def forward(self, *a, **k):
        raise RuntimeError("sub-modules of the engine-backed VLT5 hold parameters only; call the model itself")


def _canonical_device(device):
    """torch.device with an explicit index for CUDA ("cuda" -> the current device): tensors report "cuda:0", and the store / model checks
    compare devices for equality."""
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return d


def to_device(t, dev, dtype=None):
    """Host -> device hand-over of a batch tensor.  A tensor in PINNED host memory is copied stream-ordered without a host wait
    (the host keeps its lead of a whole step over the device: a blocking copy queues behind the previous step's kernels and
    then leaves the device idle while the first launches of the new step are enqueued); pageable memory blocks as in the
    reference's `.to(device)`.  The caller must not rewrite a pinned batch tensor before the step that consumes it has run."""
    nb = t.device.type == "cpu" and t.is_pinned()
    return t.to(device=dev, dtype=t.dtype if dtype is None else dtype, non_blocking=nb).contiguous()


class VLOutput(OrderedDict):
    """ModelOutput-like record (VLSeq2SeqLMOutput, modeling_t5_our.py:774-833): key and attribute access."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


class _EngineStep(torch.autograd.Function):
    """The whole forward/backward as ONE autograd node: forward returns the (per-token or reduced) loss, backward
    runs the engine's backward and deposits the gradients in `param.grad` (views of the flat gradient buffer)."""

    @staticmethod
    def forward(ctx, anchor, model, state, fused):
        ctx.model, ctx.state, ctx.fused = model, state, fused
        return state["loss"].clone() if fused else state["loss_tok"].clone()

    @staticmethod
    def backward(ctx, g):
        ctx.model._engine_backward(ctx.state, g.contiguous(), ctx.fused)
        return None, None, None, None


class VLT5(nn.Module):
    def __init__(self, config, device=None):
        super().__init__()
        self.config = config if hasattr(config, "vocab_size") else VLT5Config.from_any(config)
        self.cfg = VLT5Config.from_any(config)
        self.model_dim = self.cfg.d_model
        self.L = 20          # constant split between "question" and "visual" tokens (modeling_t5_our.py:381)
        self.V_L = 36
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self._device = _canonical_device(device)
        self.tokenizer = None
        self._ws = None
        self._lut_cache = {}
        self._step_count = 0
        self.base_seed = 0x5EED
        self.dp = None                       # set by parallel.DataParallelVLT5
        self.tuning = L.tuning_from_env()       # experiment switches (vlt5_tuning): the environment is read HERE, once, never in the library
        self.side_stream_enabled = os.environ.get("VQACL_SIDE_STREAM", "0") == "1"    # weight gradients on a second stream (measured: no gain)
        self._side = None
        self._side_events = None
        self.external_bf16_sync = False      # True once a fused optimizer keeps the bf16 shadow fresh itself
        self._gnorm = None                   # per-tile sums of squares left by the weight-gradient GEMMs (vlt5_step.gnorm_partials)
        self._gnorm_version = None           # version of the flat gradient buffer they describe (None: not valid)
        self._opt_events = None              # per-bucket events of an overlapped optimizer update (FusedAdamW(overlap=True))
        self._bf16_version = -1
        self._build(self.cfg, None)
        self.proto = PrototypeHead(self.cfg.n_ques, self.cfg.n_cate, self.cfg.d_model, self._device)
        self.init_weights()

    # ------------------------------------------------------------------ construction ----------------
    @classmethod
    def from_pretrained(cls, name=None, config=None, **kw):
        """The reference calls `model_class.from_pretrained('t5-base', config=config)` and then re-initialises every
        weight (`--from_scratch`, trainer_base.py:218-238); no checkpoint exists offline, so this only builds the model."""
        return cls(config, **kw)

    def _build(self, cfg, old_values):
        layout, total, nbuckets = param_layout(cfg)
        dev = self._device
        self._layout, self._total, self._nbuckets = layout, total, nbuckets
        self._flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self._flat_bf16 = torch.zeros(total, device=dev, dtype=torch.bfloat16)
        self._flat_grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self._flat_grad_tmp = None
        self._views, self._gviews, self._pinfo = {}, {}, {}
        for name, off, shape, bucket, decay, used in layout:
            n = int(np.prod(shape))
            self._views[name] = self._flat[off:off + n].view(*shape)
            self._gviews[name] = self._flat_grad[off:off + n].view(*shape)
            self._pinfo[name] = (off, n, bucket, decay, used)
        if old_values:
            with torch.no_grad():
                for k, v in old_values.items():
                    if k in self._views:
                        tgt = self._views[k]
                        sl = tuple(slice(0, min(a, b)) for a, b in zip(tgt.shape, v.shape))
                        tgt[sl].copy_(v[sl])
        # module tree with the reference's names; leaves are nn.Linear / nn.Embedding / norm holders so that
        # `model.apply(init_bert_weights)` (trainer_base.py:227-237) touches exactly the same tensors
        for child in list(self._modules):
            del self._modules[child]
        params = {name: nn.Parameter(v, requires_grad=True) for name, v in self._views.items()}
        self._params_by_name = params
        shared = params["shared.weight"]
        self.shared = self._leaf("shared.weight", params)
        self._attach("encoder.embed_tokens", self.shared)
        self._attach("decoder.embed_tokens", self.shared)
        for name in params:
            if name == "shared.weight":
                continue
            mod_path, leaf = name.rsplit(".", 1)
            node = self._ensure(mod_path, params)
            if leaf not in node._parameters or node._parameters[leaf] is not params[name]:
                node._parameters[leaf] = params[name]
        self._attach("encoder.visual_embedding.obj_order_embedding", self.shared)
        lm = _Linear(1, 1, bias=False)
        lm.in_features, lm.out_features = cfg.d_model, cfg.vocab_size
        lm.weight = shared                    # tied (modeling_t5_our.py:661 asserts tie_word_embeddings)
        self.lm_head = lm
        self._bf16_version = -1

    def _leaf(self, name, params):
        p = params[name]
        if name.endswith("embedding.weight") or name == "shared.weight" or "relative_attention_bias" in name:
            m = _Embedding(1, 1)
            m.num_embeddings, m.embedding_dim = p.shape
            m.weight = p
        elif p.dim() == 2:
            m = _Linear(1, 1, bias=False)
            m.out_features, m.in_features = p.shape
            m.weight = p
        else:
            m = _Holder()
            m.weight = p
        return m

    def _ensure(self, mod_path, params):
        """Create (or fetch) the module at dotted path `mod_path`; its kind is derived from its weight's name."""
        node = self
        parts = mod_path.split(".")
        for i, part in enumerate(parts):
            if part not in node._modules:
                full = ".".join(parts[:i + 1])
                wname = full + ".weight"
                if i == len(parts) - 1 and wname in params:
                    child = self._leaf(wname, params)
                    if full + ".bias" in params:
                        child.bias = params[full + ".bias"]
                else:
                    child = _Holder()
                node.add_module(part, child)
            node = node._modules[part]
        return node

    def _attach(self, mod_path, module):
        parent_path, leaf = mod_path.rsplit(".", 1)
        node = self
        for part in parent_path.split("."):
            if part not in node._modules:
                node.add_module(part, _Holder())
            node = node._modules[part]
        node.add_module(leaf, module)

    def _apply(self, fn, recurse=True):
        """`.to(device)` / `.cuda()`: move the flat buffers and re-point the parameter views (a plain per-parameter
        move would silently break the flat layout the engine relies on)."""
        probe = fn(torch.empty(0, device=self._flat.device, dtype=torch.float32))
        if probe.dtype != torch.float32:
            raise L.Vlt5Error("master parameters stay fp32; bf16 compute is internal to the engine")
        if probe.device != self._flat.device:
            values = {k: v.detach().clone() for k, v in self._views.items()}
            self._device = probe.device
            self._build(self.cfg, {k: fn(v) for k, v in values.items()})
            old = self.proto
            self.proto = PrototypeHead(self.cfg.n_ques, self.cfg.n_cate, self.cfg.d_model, self._device)
            self.proto.Q_prototype.copy_(old.Q_prototype)
            self.proto.V_prototype.copy_(old.V_prototype)
            self._ws = None
            self._lut_cache = {}
        return self

    # ------------------------------------------------------------------ reference-surface helpers ----
    def resize_token_embeddings(self, new_num_tokens=None):
        if new_num_tokens is None or new_num_tokens == self.cfg.vocab_size:
            return self.shared
        old = {k: v.detach().clone() for k, v in self._views.items()}
        self.cfg.vocab_size = new_num_tokens
        if hasattr(self.config, "vocab_size"):
            self.config.vocab_size = new_num_tokens
        self._build(self.cfg, old)
        if new_num_tokens > old["shared.weight"].shape[0]:
            with torch.no_grad():
                self._views["shared.weight"][old["shared.weight"].shape[0]:].normal_(0.0, 1.0)
        return self.shared

    @torch.no_grad()
    def init_weights(self):
        """Distributions of the reference's from-scratch init (trainer_base.py:218-238 followed by the T5
        `_init_weights`): see SURVEY 8a-18.  The RNG draw order is ours."""
        d, cfg = self.cfg.d_model, self.cfg
        inner = cfg.num_heads * cfg.d_kv
        for name, v in self._views.items():
            if name.endswith("layer_norm.weight") or name.endswith("embedding.1.weight"):
                v.fill_(1.0)
            elif name.endswith(".bias"):
                v.zero_()
            else:
                std = 1.0
                if "Attention.q." in name:
                    std = (d * cfg.d_kv) ** -0.5
                elif "Attention.k." in name or "Attention.v." in name or "relative_attention_bias" in name:
                    std = d ** -0.5
                elif "Attention.o." in name:
                    std = inner ** -0.5
                elif ".wi." in name or ".wi_0." in name or ".wi_1." in name:
                    std = d ** -0.5
                elif ".wo." in name:
                    std = cfg.d_ff ** -0.5
                v.normal_(0.0, std)
        self._bf16_version = -1

    def get_input_embeddings(self):
        return self.shared

    @property
    def Q_prototype(self):
        return self.proto.Q_prototype

    @Q_prototype.setter
    def Q_prototype(self, v):
        self.proto.Q_prototype.copy_(v.to(self.proto.Q_prototype.device))

    @property
    def V_prototype(self):
        return self.proto.V_prototype

    @V_prototype.setter
    def V_prototype(self, v):
        self.proto.V_prototype.copy_(v.to(self.proto.V_prototype.device))

    # ------------------------------------------------------------------ engine plumbing ---------------
    def _require_whole_master(self, what):
        """ZeRO-1 data parallel with gather_master=False: the f32 master of the layer buckets is only current on the owning rank.
        Reading parameters then needs the collective `dp.consolidate()` first -- a silent collective here would hang the
        reference's rank-0-only checkpoint (vqacl.py:413-414), so this raises instead."""
        if self.dp is not None and getattr(self.dp, "params_sharded", False):
            raise L.Vlt5Error(f"{what}: the f32 master weights are sharded over the data-parallel ranks "
                              "(DataParallelVLT5(algo='zero1', gather_master=False)); call dp.consolidate() on EVERY rank first, "
                              "or keep the default gather_master=True")

    def flat_params(self):
        self.sync_optimizer()
        self._require_whole_master("flat_params()")
        return self._flat

    def flat_grads(self, collective=False):
        """The flat f32 gradient buffer `.grad` views.  Data parallel: the averaged gradients may still lie in the bf16 staging
        buffer (a local cast), or -- zero1, between backward and the optimizer step -- only this rank's chunks are reduced and the
        rest has to be all-gathered: a COLLECTIVE every rank must enter.  That case raises unless the caller says
        `collective=True` (a rank-0-only gradient-norm log would otherwise hang the job)."""
        dp = self.dp
        if dp is not None and getattr(dp, "shards_valid", False) and getattr(dp, "world", 1) > 1 and not collective:
            raise L.Vlt5Error("flat_grads(): under DataParallelVLT5(algo='zero1') the gradients are reduce-scattered between backward "
                              "and the optimizer step; completing them is a collective -- call flat_grads(collective=True) on EVERY rank")
        if dp is not None and (getattr(dp, "g16_valid", False) or getattr(dp, "shards_valid", False)):
            dp.materialize_grads(self)
        return self._flat_grad

    def flat_bf16(self):
        return self._flat_bf16

    def sync_bf16(self, force=False):
        """Refresh the bf16 shadow when the fp32 master changed.  Every torch-side in-place write (load_state_dict,
        load_checkpoint, `p.data.copy_`, a stock optimizer) bumps the flat buffer's version counter; the fused optimizer
        writes both buffers through raw pointers (no bump) and records the version it left behind, so a plain version
        compare is exact in both cases."""
        if force or self._flat._version != self._bf16_version or self._bf16_version < 0:
            ops.cast_bf16(self._flat, self._flat_bf16)
            self._bf16_version = self._flat._version

    def load_state_dict(self, state_dict, strict=True, **kw):
        """nn.Module.load_state_dict + a forced refresh of the bf16 shadow (the engine reads the shadow, not the master)."""
        self.sync_optimizer()
        res = super().load_state_dict(state_dict, strict=strict, **kw)
        self._bf16_version = -1
        if self.dp is not None and getattr(self.dp, "params_sharded", False):
            # zero1 with gather_master=False: a state dict that carries every trained tensor overwrote the whole f32 master on this
            # rank, so parameter reads are local again (the Adam moments stay sharded: they belong to the owning rank either way)
            missing = {k for k in getattr(res, "missing_keys", ()) if self._pinfo.get(k, (0, 0, 0, 0, 0))[4]}
            if not missing:
                self.dp.params_sharded = False
        return res

    def _lut(self, q, k, bidirectional):
        key = (q, k, bidirectional)
        if key not in self._lut_cache:
            t = bucket_table(q, k, bidirectional, self.cfg.relative_attention_num_buckets, 128)
            self._lut_cache[key] = torch.from_numpy(np.ascontiguousarray(t)).to(self._device)
        return self._lut_cache[key]

    def _workspace(self, B, Lt, V, T):
        c = self.cfg.c_struct()
        need = lib().vlt5_workspace_bytes(C.byref(c), B, Lt, V, T)
        if need <= 0:
            raise L.Vlt5Error("vlt5_workspace_bytes rejected the shape")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need * 1.05) + 4096, device=self._device, dtype=torch.uint8)
        return self._ws

    def _ws_view(self, c, dims, which, dtype, shape):
        off = lib().vlt5_workspace_offset(C.byref(c), *dims, which)
        n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        return self._ws[off:off + n].view(dtype).view(*shape)

    def _make_step(self, st, grads=None):
        s = CStep()
        s.B, s.L, s.V, s.T = st["dims"]
        s.training, s.seed = int(st["training"]), st["seed"]
        s.params, s.params_bf16 = ptr(self._flat), ptr(self._flat_bf16)
        s.grads = ptr(grads)
        s.workspace, s.workspace_bytes = ptr(self._ws), self._ws.numel()
        s.vis_feats, s.boxes, s.input_ids = ptr(st["feats"]), ptr(st["boxes"]), ptr(st["input_ids"])
        s.labels, s.scores = ptr(st["labels"]), ptr(st.get("scores"))
        ref = st.get("feat_ref")
        if ref is not None:                 # the step's visual inputs are rows of the resident feature store (vqacl_amd/feed.py)
            s.feat_store, s.box_store = ptr(ref.store.feats), ptr(ref.store.boxes)
            s.feat_slots, s.n_slots = ptr(ref.slots), ref.store.capacity
        s.enc_lut, s.dec_lut = ptr(st["enc_lut"]), ptr(st["dec_lut"])
        s.tuning = C.pointer(self.tuning)
        if self._opt_events is not None:
            # an optimizer is (possibly still) updating the parameters on its own stream: the engine waits bucket by bucket
            arr = (L.vp * len(self._opt_events))(*[L.vp(e.cuda_event) for e in self._opt_events])
            st["_wait_arr"] = arr                                   # keep the array alive as long as the step state
            s.wait_events, s.n_wait_events = arr, len(self._opt_events)
        return s

    def _visual_inputs(self, vis_inputs):
        """(feats f32, boxes f32, V, None) for the reference's `(vis_feats, boxes)` pair, or (None, None, V, ref) for a
        `feed.StoreRef` -- rows of the HBM-resident feature store, gathered by the engine itself."""
        from .feed import StoreRef
        dev = self._device
        if isinstance(vis_inputs, StoreRef):
            st = vis_inputs.store
            if st.feats.device != dev or st.feat_dim != self.cfg.feat_dim:
                raise L.Vlt5Error("feature store lives on another device or has another feature width than the model")
            if vis_inputs.slots.device != dev or vis_inputs.slots.dtype != torch.long:
                raise L.Vlt5Error("StoreRef.slots must be an int64 tensor on the model's device")
            return None, None, st.V, vis_inputs
        feats = to_device(vis_inputs[0], dev, torch.float32)
        boxes = to_device(vis_inputs[1], dev, torch.float32)
        return feats, boxes, feats.shape[1], None

    def sync_optimizer(self):
        """Make the current stream wait for an overlapped optimizer update (FusedAdamW(overlap=True)) -- call before
        touching parameters with anything but train_step/test_step (state_dict() does it itself)."""
        if self._opt_events is not None:
            cur = torch.cuda.current_stream()
            for e in self._opt_events:
                cur.wait_event(e)
        ready = getattr(self.dp, "master_ready", None) if self.dp is not None else None
        if ready is not None:           # ZeRO-1 data parallel: the f32 master chunks of the other ranks are (still) arriving
            torch.cuda.current_stream().wait_event(ready)

    def state_dict(self, *a, **k):
        """Local on every rank, under every data-parallel mode (the reference saves on rank 0 only: trainer_base.py:246-249)."""
        self.sync_optimizer()
        self._require_whole_master("state_dict()")
        return super().state_dict(*a, **k)

    # ------------------------------------------------------------------ forward ----------------------
    def forward(self, input_ids=None, vis_inputs=None, labels=None, decoder_input_ids=None, cate_labels=None,
                ques_labels=None, proto_update=False, memory=False, current_task_id=0, proto_alpha=0.5, proto_beta=0.3,
                return_dict=True, reduce_loss=False, scores=None, _reuse_encoder=False, _alias_workspace=False, **kwargs):
        """Same keyword surface as the reference `VLT5.forward` (modeling_t5_our.py:514-713).  `scores` (optional,
        ours) fuses the train_step reduction of vqa_model.py:46-54 into the engine: the output then also carries the
        reduced scalar under 'loss_reduced'.

        The returned record has the fields of `VLSeq2SeqLMOutput` (modeling_t5_our.py:695-713, 774-833).  Its tensors are
        OWNED copies, as in the reference; `_alias_workspace=True` (used by `train_step`, which only hands two small tensors
        on) returns views of the engine's workspace instead, valid until the next forward."""
        if not self._flat.is_cuda:
            raise L.Vlt5Error("the engine-backed VLT5 needs a GPU (no CPU fallback)")
        if labels is None:
            raise NotImplementedError("decoding without labels is the generate path (vqa_model.test_step)")
        dev = self._device
        feats, boxes, V, ref = self._visual_inputs(vis_inputs)
        input_ids = to_device(input_ids, dev)
        labels = to_device(labels, dev)
        B, Lt = input_ids.shape
        T = labels.shape[1]
        S, Sx, d = Lt + V, Lt + V + 2, self.cfg.d_model
        dims = (B, Lt, V, T)
        self._workspace(*dims)
        self.sync_bf16()
        training = self.training
        self._step_count += 1
        st = dict(dims=dims, training=training, seed=(self.base_seed + 0x9E3779B1 * self._step_count) & 0xFFFFFFFF,
                  feats=feats, boxes=boxes, feat_ref=ref, input_ids=input_ids, labels=labels,
                  enc_lut=self._lut(Lt, Lt, True), dec_lut=self._lut(T, T, False))
        if scores is not None:
            st["scores"] = to_device(scores, dev, torch.float32)
        c = self.cfg.c_struct()
        cs = self._make_step(st)
        stream = stream_ptr()
        enc_f32 = self._ws_view(c, dims, L.WS_ENC_OUT, torch.float32, (B, Sx, d))
        enc_b16 = self._ws_view(c, dims, L.WS_ENC_EXT, torch.bfloat16, (B, Sx, d))
        loss_mem_Q = loss_mem_V = 0
        if _reuse_encoder:
            # greedy decoding: the encoder output (and the retrieved prototype rows) of this batch are still in the workspace;
            # the encoder-side layout of the arena does not depend on T (tests/test_host_cpu.py)
            idxQ, idxV = self._cached_idx
        else:
            check(lib().vlt5_encoder_fwd(C.byref(c), C.byref(cs), stream), "vlt5_encoder_fwd")
            # SS/SI prototype head (modeling_t5_our.py:583-615)
            fused_head = not self.proto.dist_enabled and not (proto_update and memory) and os.environ.get("VQACL_FUSED_HEAD", "1") != "0"
            dist_head = (self.proto.dist_enabled and proto_update and not memory and enc_f32.is_cuda
                         and os.environ.get("VQACL_FUSED_HEAD", "1") != "0")
            if dist_head:           # data parallel: the same head in two halves around one all-reduce of the class statistics (four launches)
                ql = to_device(ques_labels, dev, torch.float32)
                cl = to_device(cate_labels, dev, torch.float32)
                poolQ, poolV, idxQ, idxV = self.proto.forward_dist(enc_f32, enc_b16, S, self.L, ql, cl, int(current_task_id),
                                                                   float(proto_alpha), float(proto_beta))
            elif fused_head:        # pooling, state update and retrieval of both heads in three launches (vlt5_proto_head_fwd)
                ql = to_device(ques_labels, dev, torch.float32) if proto_update else None
                cl = to_device(cate_labels, dev, torch.float32) if proto_update else None
                poolQ, poolV, idxQ, idxV = self.proto.forward(enc_f32, enc_b16, S, self.L, ql, cl, int(current_task_id), float(proto_alpha),
                                                               float(proto_beta), update=bool(proto_update))
            else:
                poolQ, poolV = ops.proto_pool(enc_f32, S, self.L)
                if proto_update:
                    ql = to_device(ques_labels, dev, torch.float32)
                    cl = to_device(cate_labels, dev, torch.float32)
                    if memory:
                        loss_mem_Q, loss_mem_V = self.proto.memory_loss(poolQ, poolV, ql, cl)
                    self.proto.update(poolQ, poolV, ql, cl, int(current_task_id), float(proto_alpha), float(proto_beta))
                idxQ, idxV = self.proto.retrieve(poolQ, poolV, enc_f32, enc_b16, S)
            self._cached_idx = (idxQ, idxV)
        check(lib().vlt5_decoder_fwd(C.byref(c), C.byref(cs), stream), "vlt5_decoder_fwd")
        st["loss_tok"] = self._ws_view(c, dims, L.WS_LOSS_TOK, torch.float32, (B * T,))
        st["loss"] = self._ws_view(c, dims, L.WS_LOSS, torch.float32, (1,))[0]
        fused = scores is not None
        anchor = self._params_by_name["shared.weight"]
        if torch.is_grad_enabled() and anchor.requires_grad:
            loss_out = _EngineStep.apply(anchor, self, st, fused)
        else:
            loss_out = (st["loss"] if fused else st["loss_tok"]).clone()
        out = VLOutput()
        if fused:
            out["loss_reduced"] = loss_out
            out["loss"] = st["loss_tok"].detach()
        else:
            out["loss"] = loss_out if not reduce_loss else loss_out.sum() / (labels != -100).sum().clamp(min=1)
        own = (lambda t: t) if _alias_workspace else (lambda t: t.clone())
        out["logits"] = own(self._ws_view(c, dims, L.WS_LOGITS, torch.float32, (B, T, self.cfg.vocab_size)))
        out["past_key_values"] = None          # (use_cache decoding is `greedy_generate` / `vlt5_decoder_step`)
        # output of the decoder stack (final norm + dropout, before the 1/sqrt(d) rescale), as `decoder_outputs.last_hidden_state`;
        # the engine keeps it as the bf16 operand of the lm_head GEMM, returned here widened to f32
        out["decoder_last_hidden_state"] = self._ws_view(c, dims, L.WS_DEC_OUT, torch.bfloat16, (B, T, d)).to(torch.float32)
        out["decoder_hidden_states"] = None    # (`output_hidden_states` is never set by the reference's generative path)
        out["encoder_hidden_states"] = own(enc_f32[:, :S])
        out["encoder_attention_mask"] = own(self._ws_view(c, dims, L.WS_ENC_MASK_EXT, torch.float32, (B, Sx)))
        out["loss_memory_Q"], out["loss_memory_V"] = loss_mem_Q, loss_mem_V
        out["max_idx_Q"], out["max_idx_V"] = idxQ, idxV
        return out

    # ------------------------------------------------------------------ backward ---------------------
    def grad_release_plan(self):
        """[(phase, lo, hi)]: the gradient-bucket ranges in the order a backward completes them -- phase 0: signalled by
        vlt5_decoder_bwd, phase 1: by vlt5_encoder_bwd (upper half of the encoder mid-phase, then embeddings + norms + visual embedding
        BEFORE the last weight-gradient GEMMs, then the lower half).  The decoder layers' buckets belong to phase 1 when their weight
        gradients ride in the encoder phase's launches (vlt5_decoder_buckets_late).  A wait for a bucket's event must be enqueued
        AFTER the call that records it; the data-parallel wrapper cuts its collectives -- and, sharded, its chunk ownership -- by this."""
        return self._release_plan()[0]

    def _release_plan(self):
        """(plan, id) as the ENGINE decides them (vlt5_grad_release_plan: the same predicates its backward phases evaluate): the host never
        re-derives the order.  The id travels back in vlt5_step.release_plan_id, so a backward whose order differs refuses to run."""
        c = self.cfg.c_struct()
        tri = (C.c_int * 15)()
        n = C.c_int(0)
        pid = lib().vlt5_grad_release_plan(C.byref(c), C.byref(self.tuning), int(self.side_stream_enabled), tri, 5, C.byref(n))
        if pid <= 0:
            raise L.Vlt5Error("vlt5_grad_release_plan failed")
        return [(tri[3 * i], tri[3 * i + 1], tri[3 * i + 2]) for i in range(n.value)], pid

    def _engine_backward(self, st, g, fused):
        c = self.cfg.c_struct()
        direct = all(p.grad is None for p in self._params_by_name.values())
        if direct:
            target = self._flat_grad
        else:
            if self.dp is not None and (getattr(self.dp, "g16_valid", False) or getattr(self.dp, "shards_valid", False)):
                # accumulate onto the averaged gradients of the previous backward, not the local / partly reduced ones.  (zero1: an
                # all-gather -- fine HERE, every rank runs this backward and its other collectives at the same point anyway)
                self.dp.materialize_grads(self)
            if self._flat_grad_tmp is None:
                self._flat_grad_tmp = torch.zeros_like(self._flat_grad)
            target = self._flat_grad_tmp
        cs = self._make_step(st, target)
        self._gnorm_version = None
        fused_norm = direct and self.dp is None and os.environ.get("VQACL_FUSED_GNORM", "1") != "0"
        if fused_norm:
            # single process, gradients written straight into the flat buffer: the weight-gradient GEMMs leave their share of
            # sum(g^2) per output tile, so FusedAdamW's clip needs no second pass over the 0.9 GB of gradients
            if self._gnorm is None:
                n = lib().vlt5_gnorm_slots(C.byref(c))
                self._gnorm = torch.zeros(int(n), device=self._device) if n > 0 else False
            if self._gnorm is not False:
                cs.gnorm_partials = ptr(self._gnorm)
            else:
                fused_norm = False
        cs.defer_decoder_wgrads = 1           # the decoder's short weight gradients ride in the encoder phase's launches (vlt5_tuning.wgrad_shadow)
        gt = g.reshape(-1).to(torch.float32).contiguous()
        if fused:
            cs.gout = ptr(gt)
        else:
            cs.d_loss_tok = ptr(gt)
        keep = (g, gt)
        events = None
        if self.dp is not None and direct:
            events = self.dp.make_events(self._nbuckets)
            # only the events the wrapper waits for (the last bucket of every merged slice) are recorded: a marker in the chain's
            # queue is not free, and 26 of them per backward bought nothing
            plan, plan_id = self._release_plan()
            self.dp.check_release_plan(plan)           # frozen at wrapper construction (chunk ownership and Adam moments follow from it)
            cs.release_plan_id = plan_id               # ... and checked again by the engine calls against what THEY will do
            need = set()
            for _, lo_, hi_ in plan:
                need.update(last for _, _, _, last in self.dp.slices_of(lo_, hi_))
            arr = (L.vp * len(events))(*[L.vp(e.cuda_event) if i in need else L.vp() for i, e in enumerate(events)])
            cs.events, cs.n_events = arr, len(events)
            keep = keep + (arr,)
        mirrored = False
        if events is not None and self.dp.mirror_enabled():
            # bf16 buckets: the weight-gradient GEMMs write the staging copy of every layer bucket themselves (no cast pass)
            cs.grads_bf16 = ptr(self.dp.staging(target))
            mirrored = True
        if self.side_stream_enabled:
            # the batched weight-gradient GEMMs run on a second stream beside the input-gradient chain; vlt5_encoder_bwd joins it
            if self._side is None:
                if os.environ.get("VQACL_SIDE_STREAM_PRIO", "low") == "normal":
                    self._side = torch.cuda.Stream(device=self._device)                                 # normal priority (round-5 A/B)
                else:
                    raw = L.vp()
                    check(lib().vlt5_side_stream_create(C.byref(raw)), "vlt5_side_stream_create")      # lowest priority
                    self._side = torch.cuda.ExternalStream(raw.value, device=self._device)
                self._side_events = [torch.cuda.Event() for _ in range(4)]
                for e in self._side_events:                 # force creation of the underlying hipEvent_t
                    e.record()
            sarr = (L.vp * 4)(*[L.vp(e.cuda_event) for e in self._side_events])
            cs.side_stream, cs.side_events, cs.n_side_events = L.vp(self._side.cuda_stream), sarr, 4
            keep = keep + (sarr,)
        stream = stream_ptr()
        check(lib().vlt5_decoder_bwd(C.byref(c), C.byref(cs), stream), "vlt5_decoder_bwd")
        if events is not None:
            # collectives are cut where the engine releases gradients (grad_release_plan); the wait for a bucket's event is enqueued
            # after the engine call that records it (a wait enqueued earlier would see the PREVIOUS backward's record)
            for phase, lo, hi in plan:
                if phase == 0:
                    self.dp.reduce_range(self, events, lo, hi, mirrored=mirrored)
        check(lib().vlt5_encoder_bwd(C.byref(c), C.byref(cs), stream), "vlt5_encoder_bwd")
        if events is not None:
            # (decoder layers when their weight gradients rode in this phase,) upper half of the encoder (mid-phase), then embeddings +
            # norms + visual embedding (released BEFORE the last weight-gradient GEMMs: their exchange hides under those; bf16
            # buckets: the engine mirrored them too -- GEMM epilogues + vlt5_mirror_rows_bf16 -- no cast pass), then the lower half
            # of the encoder -- the only group whose exchange stays exposed
            late = [(lo, hi) for phase, lo, hi in plan if phase == 1]
            for i, (lo, hi) in enumerate(late):
                self.dp.reduce_range(self, events, lo, hi, final=(i == len(late) - 1), mirrored=mirrored)
            self.dp.finish()
        elif self.dp is not None:
            self.dp.reduce_flat(target)
        for name, p in self._params_by_name.items():
            off, n, bucket, decay, used = self._pinfo[name]
            if not used:
                continue                      # prototype_fc1/2 never receive a gradient
            if direct:
                p.grad = self._gviews[name]
            else:
                gv = target[off:off + n].view(p.shape)
                if p.grad is None:
                    p.grad = gv.clone()
                else:
                    p.grad.add_(gv)
        if fused_norm:
            self._gnorm_version = self._flat_grad._version     # any later in-place edit of a .grad view bumps it: the shares go stale
        del keep
