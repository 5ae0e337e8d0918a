"""VQA task wrapper = the drop-in boundary (reference: VL-T5/src/vqa_model.py:10-121).

`Trainer.train_step` (src/vqacl.py:438) calls `model.module.train_step(batch, task_idx, proto_alpha, proto_beta,
each_memory, task_total_num)` and then `.backward()` on `result['loss']`; both behave exactly as in the
reference, only the arithmetic runs in the HIP engine.
"""
import torch

from .modeling_vlt5 import VLT5


class VLT5VQA(VLT5):
    def __init__(self, config, num_answers=None, label2ans=None, **kw):
        super().__init__(config, **kw)
        self.num_answers = num_answers
        self.label2ans = label2ans

    def train_step(self, batch, current_task_id, proto_alpha, proto_beta, mem_num_Q=0, total_num_Q=1000, memory=False):
        device = self._device
        lm_labels = batch["target_ids"].to(device)
        output = self(
            input_ids=batch["input_ids"],
            vis_inputs=(batch["vis_feats"], batch["boxes"]),
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
        )
        assert "loss" in output
        B, Lt = lm_labels.size()
        result = {"loss": output["loss_reduced"]}
        result["encoder_hidden_states"] = output["encoder_hidden_states"]
        result["BL"] = (B, Lt)
        result["encoder_attention_mask"] = output["encoder_attention_mask"]
        return result

    @torch.no_grad()
    def test_step(self, batch, **kwargs):
        """Greedy decoding (the reference forwards no generation kwargs, so `--num_beams` is ignored: vqa_model.py:112-116).
        The encoder and the prototype retrieval run once; each step re-decodes the growing prefix through the training
        decoder kernels (O(T^2) decoder work, T <= 20, no KV cache yet)."""
        self.eval()
        token_ids = self.greedy_generate(batch["input_ids"], (batch["vis_feats"], batch["boxes"]),
                                         max_length=kwargs.get("max_length", 20))
        result = {"token_ids": token_ids}
        if self.tokenizer is not None:
            result["pred_ans"] = self.tokenizer.batch_decode(token_ids, skip_special_tokens=True)
        return result

    @torch.no_grad()
    def greedy_generate(self, input_ids, vis_inputs, max_length=20, eos_token_id=1):
        device = self._device
        B = input_ids.shape[0]
        pad, start = self.cfg.pad_token_id, self.cfg.decoder_start_token_id
        tokens = torch.full((B, 1), start, dtype=torch.long, device=device)
        done = torch.zeros(B, dtype=torch.bool, device=device)
        input_ids = input_ids.to(device)
        vis_inputs = (vis_inputs[0].to(device), vis_inputs[1].to(device))
        self._workspace(B, input_ids.shape[1], vis_inputs[0].shape[1], max_length)    # size the arena once for the longest prefix
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
