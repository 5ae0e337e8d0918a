"""VQA task wrapper = the drop-in boundary (reference: VL-T5/src/vqa_model.py:10-121).

`Trainer.train_step` (src/vqacl.py:438) calls `model.module.train_step(batch, task_idx, proto_alpha, proto_beta,
each_memory, task_total_num)` and then `.backward()` on `result['loss']`; both behave exactly as in the
reference, only the arithmetic runs in the HIP engine.
"""
import torch

from .modeling_vlt5 import VLT5, to_device


def _vis_inputs(batch):
    """`(vis_feats, boxes)` as collated by the reference (vqa_data_memory.py:365-396), or the batch's rows of the HBM-resident
    feature store (`feat_ref`, vqacl_amd/feed.py) when the feed keeps the features on the GPU."""
    ref = batch.get("feat_ref")
    return ref if ref is not None else (batch["vis_feats"], batch["boxes"])


class VLT5VQA(VLT5):
    def __init__(self, config, num_answers=None, label2ans=None, **kw):
        super().__init__(config, **kw)
        self.num_answers = num_answers
        self.label2ans = label2ans

    def train_step(self, batch, current_task_id, proto_alpha, proto_beta, mem_num_Q=0, total_num_Q=1000, memory=False):
        device = self._device
        lm_labels = to_device(batch["target_ids"], device)
        output = self(
            input_ids=batch["input_ids"],
            vis_inputs=_vis_inputs(batch),
            labels=lm_labels,
            cate_labels=batch["cate_labels"],
            ques_labels=batch["ques_labels"],
            proto_update=True,
            memory=memory,
            current_task_id=current_task_id,
            mem_num_Q=mem_num_Q,
            total_num_Q=total_num_Q,
            proto_alpha=proto_alpha,
            proto_beta=proto_beta,
            return_dict=True,
            scores=batch["scores"],          # fuses the reduction of vqa_model.py:46-54 into the engine
            _alias_workspace=True,           # no 51 MB copy of the logits per step: only the two small tensors below are handed on
        )
        assert "loss" in output
        B, Lt = lm_labels.size()
        result = {"loss": output["loss_reduced"]}
        result["encoder_hidden_states"] = output["encoder_hidden_states"].clone()      # owned, like the reference's (vqa_model.py:58)
        result["BL"] = (B, Lt)
        result["encoder_attention_mask"] = output["encoder_attention_mask"].clone()
        return result

    @torch.no_grad()
    def test_step(self, batch, **kwargs):
        """Greedy decoding (the reference forwards no generation kwargs, so `--num_beams` is ignored: vqa_model.py:112-116).
        The encoder and the prototype retrieval run once; every further token is ONE incremental decoder step over a
        key/value cache (`vlt5_decoder_step`)."""
        from ._lib import Vlt5Error
        if getattr(self.cfg, "classifier", False):
            # vqa_model.py:81-108 applies `self.answer_head` to the last decoder state -- a module the reference never DEFINES
            # (the only two mentions in its tree are the calls at vqa_model.py:102 and nextqa/vqa_model_nextqa.py:103), so the
            # branch raises AttributeError there; no launch script passes `--classifier`.  Same outcome here, said plainly.
            raise NotImplementedError("config.classifier=True: the reference's own branch calls an undefined `answer_head` "
                                      "(vqa_model.py:102); its scripts use the generative path")
        # the reference forwards **kwargs to HF `generate`; the engine decodes greedily and says so instead of dropping options
        allowed = {"max_length", "num_beams", "eos_token_id", "use_cache", "do_sample", "early_stopping"}
        unknown = set(kwargs) - allowed
        if unknown:
            raise Vlt5Error(f"test_step: unsupported generation arguments {sorted(unknown)} (greedy decoding only)")
        if kwargs.get("num_beams") not in (None, 1) or kwargs.get("do_sample"):
            raise Vlt5Error("test_step: only greedy decoding is implemented (num_beams=1, do_sample=False) -- the reference's "
                            "Trainer.predict passes no generation arguments either (vqacl.py:594)")
        self.eval()
        token_ids = self.greedy_generate(batch["input_ids"], _vis_inputs(batch), max_length=kwargs.get("max_length", 20),
                                         eos_token_id=kwargs.get("eos_token_id", 1), use_cache=kwargs.get("use_cache", True))
        result = {"token_ids": token_ids}
        if self.tokenizer is not None:
            result["pred_ans"] = self.tokenizer.batch_decode(token_ids, skip_special_tokens=True)
        return result

    @torch.no_grad()
    def greedy_generate(self, input_ids, vis_inputs, max_length=20, eos_token_id=1, use_cache=True):
        """HF `generate` semantics for greedy search: starts from decoder_start_token_id, a finished row keeps emitting pad,
        stops when every row has produced EOS or at max_length.  use_cache=False re-decodes the growing prefix through the
        training decoder kernels (O(T^2)); kept as the cross-check of the cached path."""
        if not use_cache:
            return self._greedy_generate_recompute(input_ids, vis_inputs, max_length, eos_token_id)
        import ctypes as C
        from . import _lib as L
        from . import ops
        from ._lib import check, lib, ptr, stream_ptr
        if self.training:
            raise L.Vlt5Error("greedy_generate runs in eval mode (call .eval() or test_step)")
        if not 2 <= max_length <= 64:
            raise L.Vlt5Error("max_length must be in [2, 64] (the key/value cache of the attention kernel)")
        dev = self._device
        feats, boxes, V, ref = self._visual_inputs(vis_inputs)
        input_ids = input_ids.to(dev).contiguous()
        B, Lt = input_ids.shape
        Tcap = int(max_length)
        S, Sx, d = Lt + V, Lt + V + 2, self.cfg.d_model
        dims = (B, Lt, V, Tcap)
        self._workspace(*dims)
        self.sync_bf16()
        pad, start = self.cfg.pad_token_id, self.cfg.decoder_start_token_id
        st = dict(dims=dims, training=False, seed=0, feats=feats, boxes=boxes, feat_ref=ref, input_ids=input_ids,
                  labels=torch.zeros(B, Tcap, dtype=torch.long, device=dev),
                  enc_lut=self._lut(Lt, Lt, True), dec_lut=self._lut(Tcap, Tcap, False))
        c = self.cfg.c_struct()
        cs = self._make_step(st)
        stream = stream_ptr()
        check(lib().vlt5_encoder_fwd(C.byref(c), C.byref(cs), stream), "vlt5_encoder_fwd")
        enc_f32 = self._ws_view(c, dims, L.WS_ENC_OUT, torch.float32, (B, Sx, d))
        enc_b16 = self._ws_view(c, dims, L.WS_ENC_EXT, torch.bfloat16, (B, Sx, d))
        poolQ, poolV = ops.proto_pool(enc_f32, S, self.L)
        self.proto.retrieve(poolQ, poolV, enc_f32, enc_b16, S)        # rows S, S+1 <- retrieved prototypes (modeling_t5_our.py:608-612)
        inner = self.cfg.num_heads * self.cfg.d_kv
        if lib().vlt5_decode_fast_supported(C.byref(c), C.byref(cs)) == 1:
            # the decode kernels: HF's loop body (argmax -> pad after EOS -> done flags -> next input row) runs on the device inside the
            # step, the step index lives on the device too (vlt5_greedy_desc.t_dev), so every step after the first is the SAME ~100
            # launches: they are captured once per shape in a HIP graph and replayed per token -- the host enqueues one graph launch per
            # token instead of ~100 kernels (0.3-0.45 ms of host time per step against 0.56 ms on the device: a slower host was the
            # bound) and looks at the done flags every 8 tokens
            ds = self._decode_state(B, Lt, V, Tcap, int(eos_token_id), int(pad))
            cache, out, done, cur, t_dev = ds["cache"], ds["out"], ds["done"], ds["cur"], ds["t_dev"]
            out.fill_(pad)
            out[:, 0] = start
            done.zero_()
            cur.fill_(start)
            t_dev.zero_()
            g = ds["desc"]
            g.tokens, g.kv_cache, g.out_tokens, g.out_ld, g.done = ptr(cur), ptr(cache), ptr(out), Tcap, ptr(done)
            g.eos_id, g.pad_id, g.t_dev = int(eos_token_id), int(pad), ptr(t_dev)
            steps = 0
            for t in range(Tcap - 1):
                g.t = t
                if t >= 2 and self.decode_graph and ds["graph"] is None:
                    # (steps 0 and 1 ran directly: every lazy per-kernel set-up has happened.)  Capture in thread-local mode: the
                    # reference's eval loaders run a pin-memory thread (vqa_data_memory.py:786) whose hipHostMalloc / hipEventQuery
                    # calls would invalidate a global-mode capture.  A capture that still fails is not an evaluation failure: this
                    # state then enqueues the step's launches every token (the same kernels, bit-identical tokens).
                    ds["graph"] = self._capture_token_step(c, cs, g)
                if t >= 2 and self.decode_graph and ds["graph"] is not False:
                    ds["graph"].replay()
                else:
                    check(lib().vlt5_decoder_step_greedy(C.byref(c), C.byref(cs), C.byref(g), stream), "vlt5_decoder_step_greedy")
                steps = t + 1
                if (t & 7) == 7 and bool(done.all()):
                    break
            out = out[:, :steps + 1].clone()
        else:
            cache = torch.empty(self.cfg.num_decoder_layers, B, Tcap, 2 * inner, device=dev, dtype=torch.bfloat16)
            logits = torch.empty(B, self.cfg.vocab_size, device=dev, dtype=torch.float32)
            nxt = torch.empty(B, dtype=torch.long, device=dev)
            cur = torch.full((B,), start, dtype=torch.long, device=dev)
            done = torch.zeros(B, dtype=torch.bool, device=dev)
            padv = torch.full((B,), pad, dtype=torch.long, device=dev)
            tokens = [cur]
            for t in range(Tcap - 1):
                check(lib().vlt5_decoder_step(C.byref(c), C.byref(cs), ptr(cur), t, ptr(cache), ptr(logits), ptr(nxt), stream),
                      "vlt5_decoder_step")
                cur = torch.where(done, padv, nxt)
                tokens.append(cur)
                done = done | (cur == eos_token_id)
                if (t & 3) == 3 and bool(done.all()):                     # one host sync every 4 tokens
                    break
            out = torch.stack(tokens, dim=1)
        # trim what was decoded after every row had finished (the reference stops at that token)
        alive = (out != eos_token_id).long().cumprod(dim=1)               # 1 until (excluding) a row's first EOS
        length = int(alive.sum(dim=1).max()) + 1                      # longest row incl. its EOS
        return out[:, :min(out.shape[1], max(length, 1))]

    def _capture_token_step(self, c, cs, g):
        """One token-step of the decode kernels captured in a HIP graph, or False when the capture failed (the caller then enqueues
        the launches every step).  The step that was being captured has NOT run in either case: graph capture records, it does not
        execute, and a failed capture leaves no work behind on the stream."""
        import ctypes as C
        import warnings
        from ._lib import check, lib, stream_ptr
        gr = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                check(lib().vlt5_decoder_step_greedy(C.byref(c), C.byref(cs), C.byref(g), stream_ptr()), "vlt5_decoder_step_greedy")
        except Exception as e:  # noqa: BLE001  (hipErrorStreamCaptureInvalidated and friends surface as RuntimeError / Vlt5Error)
            warnings.warn(f"greedy_generate: HIP-graph capture of the token-step failed ({type(e).__name__}: {e}); "
                          "this shape decodes with enqueued launches instead")
            torch.cuda.synchronize()
            return False
        return gr

    decode_graph = True            # replay the token-step of the decode kernels from a HIP graph (False: enqueue its launches every step)

    def _decode_state(self, B, Lt, V, Tcap, eos, pad):
        """Buffers of the greedy loop (key/value cache, emitted tokens, done flags, current input ids, the device-side step index: one set
        per (B, max_length)) and the captured graph of one token-step per shape -- the evaluator pads the questions of a batch to the
        batch's longest, so a handful of question lengths alternate: up to 16 graphs are kept.  A graph holds raw pointers: it is keyed
        by everything they come from (its buffers, the workspace arena, the parameter buffers, the engine switches) and dropped when any
        of them changed."""
        from . import _lib as L
        bufs = getattr(self, "_decode_bufs", None)
        if bufs is None:
            bufs, self._decode_states = {}, {}
            self._decode_bufs = bufs
        bk = (B, Tcap)
        if bk not in bufs:
            if len(bufs) >= 4:
                old = next(iter(bufs))
                bufs.pop(old)
                for k in [k for k in self._decode_states if (k[0], k[3]) == old]:
                    self._decode_states.pop(k)
            dev, inner = self._device, self.cfg.num_heads * self.cfg.d_kv
            bufs[bk] = dict(cache=torch.empty(self.cfg.num_decoder_layers, B, Tcap, 2 * inner, device=dev, dtype=torch.bfloat16),
                            out=torch.empty(B, Tcap, dtype=torch.long, device=dev), done=torch.empty(B, dtype=torch.int32, device=dev),
                            cur=torch.empty(B, dtype=torch.long, device=dev), t_dev=torch.zeros(1, dtype=torch.int32, device=dev))
        key = (B, Lt, V, Tcap, eos, pad)
        sig = (self._ws.data_ptr(), self._ws.numel(), self._flat.data_ptr(), self._flat_bf16.data_ptr(), bytes(self.tuning))
        ds = self._decode_states.get(key)
        if ds is None:
            if len(self._decode_states) >= 16:
                self._decode_states.pop(next(iter(self._decode_states)))
            ds = dict(bufs[bk], desc=L.GreedyDesc(), graph=None, sig=sig)
            self._decode_states[key] = ds
        if ds["sig"] != sig:
            ds["graph"], ds["sig"] = None, sig
        return ds

    @torch.no_grad()
    def _greedy_generate_recompute(self, input_ids, vis_inputs, max_length=20, eos_token_id=1):
        device = self._device
        B = input_ids.shape[0]
        pad, start = self.cfg.pad_token_id, self.cfg.decoder_start_token_id
        tokens = torch.full((B, 1), start, dtype=torch.long, device=device)
        done = torch.zeros(B, dtype=torch.bool, device=device)
        input_ids = input_ids.to(device)
        self._workspace(B, input_ids.shape[1], self._visual_inputs(vis_inputs)[2], max_length)    # size the arena once for the longest prefix
        for step in range(max_length - 1):
            # labels whose shift-right equals the current prefix: prefix[1:] followed by one dummy position
            labels = torch.cat([tokens[:, 1:], torch.full((B, 1), pad, dtype=torch.long, device=device)], dim=1)
            # the encoder runs once; later steps only re-run the decoder on the grown prefix
            out = self(input_ids=input_ids, vis_inputs=vis_inputs, labels=labels, proto_update=False, _reuse_encoder=step > 0)
            nxt = out["logits"][:, -1, :].argmax(dim=-1)
            nxt = torch.where(done, torch.full_like(nxt, pad), nxt)
            tokens = torch.cat([tokens, nxt[:, None]], dim=1)
            done = done | (nxt == eos_token_id)
            if bool(done.all()):
                break
        return tokens
