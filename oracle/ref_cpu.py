"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the VQACL VL-T5 hot path.

This file restates, in plain torch-CPU fp32 ops, the arithmetic of the reference's
forward/backward path so the HIP kernels in `vqacl_amd/csrc` can be checked against it.
It is NOT part of the product: `vqacl_amd/` never imports it.

What it follows (reference file:line, relative to /root/reference/VL-T5):
  * visual embedding ............ src/modeling_t5_our.py:27-143   (`visual_embedding`)
  * joint encoder assembly ...... src/modeling_t5_our.py:175-339  (`encoder_forward`)
  * prototype head .............. src/modeling_t5_our.py:434-511  (`PrototypeState`)
  * model forward ............... src/modeling_t5_our.py:514-713  (`vlt5_forward`)
  * memory loss ................. nextqa/modeling_t5_nextqa.py:544-555 (`memory_loss`)
  * train_step loss reduction ... src/vqa_model.py:18-65          (`train_step_loss`)
  * optimizer step .............. src/vqacl.py:461-487, src/trainer_base.py:130-198
  * weight init ................. src/trainer_base.py:218-238 + HF T5 `_init_weights`

Third-party arithmetic: the reference pins `transformers==4.2.1` (requirements.txt:2) whose
source is not in /root/reference.  Its T5 leaf arithmetic is restated here from the published
algorithm (RMS LayerNorm without mean/bias; bias-free q/k/v/o; UN-scaled q.k^T; bucketed
relative position bias; softmax in fp32; ReLU or gated-GELU FFN; shift-right; additive masks
-1e4 for self-attention / -1e9 for cross-attention in fp32).

Pinning (see oracle/make_golden.py, tests/golden/): the restatement is checked against
  - the reference's own `VisualEmbedding`, `VLT5.cosine_similarity_multi`,
    `VLT5.update_prototype`, `VLT5.calculate_current_prototype` and nextqa `memory_loss`
    run in the build container, and
  - the container's transformers-5.15 T5 leaf modules (T5LayerNorm, T5Attention, T5LayerFF,
    T5Block, decoder T5Stack, `_relative_position_bucket`, `_shift_right`) standing in for 4.2.1.
The reference has no tests and its full model cannot be instantiated here,
so the end-to-end composition is pinned only through those parts.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class Cfg:
    """Hyper-parameters that reach the hot path (t5-base values by default)."""
    d_model: int = 768
    d_kv: int = 64
    num_heads: int = 12
    d_ff: int = 3072
    num_layers: int = 12
    num_decoder_layers: int = 12
    vocab_size: int = 32200          # 32000 + 100 extra ids + 100 vis extra ids (tokenization.py:59-60)
    rel_buckets: int = 32
    rel_max_distance: int = 128
    eps: float = 1e-6
    dropout: float = 0.1
    feat_dim: int = 2048
    pos_dim: int = 4
    n_images: int = 2
    gated_act: bool = False          # t5-base is plain ReLU
    pad_token_id: int = 0
    decoder_start_token_id: int = 0
    n_ques: int = 10                 # question-type tasks (Question_type.py:16)
    n_cate: int = 80                 # object categories
    split_L: int = 20                # `self.L = 20`, modeling_t5_our.py:381

    @property
    def inner(self) -> int:
        return self.num_heads * self.d_kv


def tiny_cfg(**kw) -> Cfg:
    """The small configuration used by fixtures."""
    base = dict(d_model=64, d_kv=16, num_heads=4, d_ff=128, num_layers=2, num_decoder_layers=2,
                vocab_size=400, feat_dim=64, dropout=0.0)
    base.update(kw)
    return Cfg(**base)


# --------------------------------------------------------------------------------------
# leaf arithmetic (HF T5 4.2.1 semantics)
# --------------------------------------------------------------------------------------
def t5_layernorm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """RMS norm: x * rsqrt(mean(x^2) + eps) * w, statistics in fp32, no mean subtraction."""
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def relative_position_bucket(rel: Tensor, bidirectional: bool, num_buckets: int = 32,
                             max_distance: int = 128) -> Tensor:
    """Integer bucket id of `rel = key_pos - query_pos` (T5 / mesh-tensorflow scheme)."""
    out = torch.zeros_like(rel)
    nb = num_buckets
    if bidirectional:
        nb //= 2
        out = out + (rel > 0).to(torch.long) * nb
        n = rel.abs()
    else:
        n = -torch.clamp(rel, max=0)
    max_exact = nb // 2
    small = n < max_exact
    # fp32 log, python-float divisor, integer scale -- same op order as the library so that the
    # truncation lands on the same side for exact powers (n == 2*max_exact).
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact)
                         * (nb - max_exact)).to(torch.long)
    large = torch.clamp(large, max=nb - 1)
    return out + torch.where(small, n, large)


def bucket_table(qlen: int, klen: int, bidirectional: bool, num_buckets: int = 32,
                 max_distance: int = 128) -> Tensor:
    q = torch.arange(qlen, dtype=torch.long)[:, None]
    k = torch.arange(klen, dtype=torch.long)[None, :]
    return relative_position_bucket(k - q, bidirectional, num_buckets, max_distance)


def compute_bias(table: Tensor, qlen: int, klen: int, bidirectional: bool, cfg: Cfg) -> Tensor:
    """table [num_buckets, H] -> bias [1, H, qlen, klen]."""
    b = bucket_table(qlen, klen, bidirectional, cfg.rel_buckets, cfg.rel_max_distance)
    return table[b].permute(2, 0, 1).unsqueeze(0)


def t5_attention(x_q: Tensor, x_kv: Tensor, wq: Tensor, wk: Tensor, wv: Tensor, wo: Tensor,
                 bias: Tensor, cfg: Cfg, p_drop: float, training: bool) -> Tensor:
    """softmax(q k^T + bias) v -> o.  No 1/sqrt(d) scaling (T5).  `bias` already holds the mask."""
    B, Tq, _ = x_q.shape
    Tk = x_kv.shape[1]
    H, dk = cfg.num_heads, cfg.d_kv
    q = F.linear(x_q, wq).view(B, Tq, H, dk).transpose(1, 2)
    k = F.linear(x_kv, wk).view(B, Tk, H, dk).transpose(1, 2)
    v = F.linear(x_kv, wv).view(B, Tk, H, dk).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(2, 3)) + bias
    probs = F.softmax(scores.float(), dim=-1).type_as(scores)
    probs = F.dropout(probs, p=p_drop, training=training)
    ctx = torch.matmul(probs, v).transpose(1, 2).contiguous().view(B, Tq, H * dk)
    return F.linear(ctx, wo)


def gelu_new(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def t5_ffn(x: Tensor, P: Dict[str, Tensor], prefix: str, cfg: Cfg, training: bool) -> Tensor:
    p = cfg.dropout
    if cfg.gated_act:
        h = gelu_new(F.linear(x, P[prefix + "wi_0.weight"])) * F.linear(x, P[prefix + "wi_1.weight"])
    else:
        h = F.relu(F.linear(x, P[prefix + "wi.weight"]))
    h = F.dropout(h, p=p, training=training)
    return F.linear(h, P[prefix + "wo.weight"])


def shift_right(labels: Tensor, cfg: Cfg) -> Tensor:
    """labels -> decoder inputs: prepend start id, drop last, -100 -> pad."""
    out = labels.new_zeros(labels.shape)
    out[..., 1:] = labels[..., :-1]
    out[..., 0] = cfg.decoder_start_token_id
    return out.masked_fill(out == -100, cfg.pad_token_id)


# --------------------------------------------------------------------------------------
# visual embedding  (modeling_t5_our.py:27-143)
# --------------------------------------------------------------------------------------
def visual_embedding(P: Dict[str, Tensor], feats: Tensor, boxes: Tensor, cfg: Cfg) -> Tensor:
    pre = "encoder.visual_embedding."
    B, N, _ = feats.shape
    f = F.linear(feats, P[pre + "feat_embedding.0.weight"], P[pre + "feat_embedding.0.bias"])
    f = t5_layernorm(f, P[pre + "feat_embedding.1.weight"], cfg.eps)
    # "area" as the reference computes it: columns read as (x1, x2, y1, y2) although the loader
    # provides (x1, y1, x2, y2)  (modeling_t5_our.py:78-90)
    area = (boxes[:, :, 3] - boxes[:, :, 2]) * (boxes[:, :, 1] - boxes[:, :, 0])
    pos5 = torch.cat([boxes, area.unsqueeze(2)], dim=2)
    a = F.linear(pos5, P[pre + "absolute_vis_pos_embedding.0.weight"],
                 P[pre + "absolute_vis_pos_embedding.0.bias"])
    a = t5_layernorm(a, P[pre + "absolute_vis_pos_embedding.1.weight"], cfg.eps)
    img = P[pre + "img_order_embedding.weight"][0].view(1, 1, -1)
    shared = P["shared.weight"]
    obj_ids = shared.shape[0] - 1 - torch.arange(N)
    obj = shared[obj_ids].unsqueeze(0)
    return f + a + img + obj


# --------------------------------------------------------------------------------------
# encoder / decoder
# --------------------------------------------------------------------------------------
def encoder_forward(P: Dict[str, Tensor], input_ids: Tensor, feats: Tensor, boxes: Tensor,
                    cfg: Cfg, training: bool) -> Tuple[Tensor, Tensor]:
    """Returns (last_hidden_state [B,S,d], attention_mask [B,S])."""
    B, L = input_ids.shape
    txt = P["shared.weight"][input_ids]
    vis = visual_embedding(P, feats, boxes, cfg)
    V = vis.shape[1]
    S = L + V
    x = torch.cat([txt, vis], dim=1)
    mask = torch.cat([(input_ids != cfg.pad_token_id).to(x.dtype), x.new_ones(B, V)], dim=1)
    ext = (1.0 - mask)[:, None, None, :] * -10000.0
    table = P["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
    bias = x.new_zeros(1, cfg.num_heads, S, S)
    bias[:, :, :L, :L] = compute_bias(table, L, L, True, cfg)      # text<->text only
    bias = bias + ext
    p = cfg.dropout
    x = F.dropout(x, p=p, training=training)
    for i in range(cfg.num_layers):
        pre = f"encoder.block.{i}.layer."
        a = pre + "0.SelfAttention."
        xn = t5_layernorm(x, P[pre + "0.layer_norm.weight"], cfg.eps)
        y = t5_attention(xn, xn, P[a + "q.weight"], P[a + "k.weight"], P[a + "v.weight"],
                         P[a + "o.weight"], bias, cfg, p, training)
        x = x + F.dropout(y, p=p, training=training)
        xn = t5_layernorm(x, P[pre + "1.layer_norm.weight"], cfg.eps)
        y = t5_ffn(xn, P, pre + "1.DenseReluDense.", cfg, training)
        x = x + F.dropout(y, p=p, training=training)
    x = t5_layernorm(x, P["encoder.final_layer_norm.weight"], cfg.eps)
    x = F.dropout(x, p=p, training=training)
    return x, mask


def decoder_forward(P: Dict[str, Tensor], dec_ids: Tensor, enc_hidden: Tensor, enc_mask: Tensor,
                    cfg: Cfg, training: bool) -> Tensor:
    B, T = dec_ids.shape
    x = P["shared.weight"][dec_ids]
    causal = torch.tril(torch.ones(T, T, dtype=x.dtype))
    self_ext = (1.0 - causal)[None, None, :, :] * -10000.0
    table = P["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
    self_bias = compute_bias(table, T, T, False, cfg) + self_ext
    cross_bias = (1.0 - enc_mask)[:, None, None, :] * -1e9
    cross_bias = cross_bias.expand(B, 1, T, enc_mask.shape[1])
    p = cfg.dropout
    x = F.dropout(x, p=p, training=training)
    for i in range(cfg.num_decoder_layers):
        pre = f"decoder.block.{i}.layer."
        a = pre + "0.SelfAttention."
        xn = t5_layernorm(x, P[pre + "0.layer_norm.weight"], cfg.eps)
        y = t5_attention(xn, xn, P[a + "q.weight"], P[a + "k.weight"], P[a + "v.weight"],
                         P[a + "o.weight"], self_bias, cfg, p, training)
        x = x + F.dropout(y, p=p, training=training)
        c = pre + "1.EncDecAttention."
        xn = t5_layernorm(x, P[pre + "1.layer_norm.weight"], cfg.eps)
        y = t5_attention(xn, enc_hidden, P[c + "q.weight"], P[c + "k.weight"], P[c + "v.weight"],
                         P[c + "o.weight"], cross_bias, cfg, p, training)
        x = x + F.dropout(y, p=p, training=training)
        xn = t5_layernorm(x, P[pre + "2.layer_norm.weight"], cfg.eps)
        y = t5_ffn(xn, P, pre + "2.DenseReluDense.", cfg, training)
        x = x + F.dropout(y, p=p, training=training)
    x = t5_layernorm(x, P["decoder.final_layer_norm.weight"], cfg.eps)
    return F.dropout(x, p=p, training=training)


# --------------------------------------------------------------------------------------
# SS/SI prototype head  (modeling_t5_our.py:434-511)
# --------------------------------------------------------------------------------------
def calculate_current_prototype(hidden_slice: Tensor, onehot: Tensor) -> Tuple[Tensor, Tensor]:
    """mean over tokens, then per-class mean over the batch; classes with no sample give 0."""
    pooled = hidden_slice.mean(dim=1)                               # [B,d]
    cnt = onehot.sum(dim=0)                                         # [C]
    div = torch.where(cnt <= 0, torch.ones_like(cnt), cnt).unsqueeze(1)
    proto = onehot.t().matmul(pooled) / div                         # [C,d]
    return proto, cnt


def cosine_retrieve(protos: Tensor, pooled: Tensor) -> Tuple[Tensor, Tensor]:
    """argmax_c cos(tanh P_c, tanh x_b); returns (P[idx] [B,d], idx [B] int64)."""
    a = F.normalize(torch.tanh(protos), dim=1)
    b = F.normalize(torch.tanh(pooled), dim=1)
    sim = F.linear(a, b).transpose(1, 0)                            # [B,C]
    idx = torch.argmax(sim, dim=1)
    return protos[idx], idx


@dataclass
class PrototypeState:
    """The per-task state machine of `VLT5.update_prototype` (modeling_t5_our.py:465-498)."""
    Q_prototype: Optional[Tensor] = None
    V_prototype: Optional[Tensor] = None
    Q_prototype_num: Optional[Tensor] = None
    V_prototype_num: Optional[Tensor] = None
    Q_task_cur_proto: Dict[int, Tensor] = field(default_factory=dict)
    Q_task_mem_proto: Dict[int, Tensor] = field(default_factory=dict)

    def update(self, cur_Q: Tensor, cur_V: Tensor, num_Q: Tensor, num_V: Tensor, task: int,
               alpha: float, beta: float) -> None:
        cur_Q, cur_V = cur_Q.detach(), cur_V.detach()
        if task not in self.Q_task_cur_proto:
            # first batch of a task: counts and the V prototypes restart from this batch
            self.Q_task_cur_proto[task] = cur_Q
            self.Q_prototype_num = num_Q
            self.V_prototype_num = num_V
            self.V_prototype = cur_V
            if task == 0:
                self.Q_prototype = cur_Q
            else:
                self.Q_prototype[task] = cur_Q[task]
            return
        self.Q_task_cur_proto[task] = cur_Q
        if task != 0:
            mem_now = cur_Q.clone()
            mem_now[task] = 0
            if task not in self.Q_task_mem_proto:
                self.Q_task_mem_proto[task] = mem_now
            else:
                self.Q_task_mem_proto[task] = alpha * self.Q_task_mem_proto[task] + (1 - alpha) * mem_now
            # same storage as the memory tensor on purpose: the reference aliases it
            self.Q_prototype = self.Q_task_mem_proto[task]
            self.Q_prototype[task] = cur_Q[task]
        else:
            self.Q_prototype = cur_Q
        self.V_prototype = beta * self.V_prototype + (1 - beta) * cur_V
        self.Q_prototype_num = self.Q_prototype_num + num_Q
        self.V_prototype_num = self.V_prototype_num + num_V


def memory_loss(hidden_Q: Tensor, hidden_V: Tensor, ques_labels: Tensor, cate_labels: Tensor,
                Q_prototype: Tensor, V_prototype: Tensor) -> Tuple[Tensor, Tensor]:
    """nextqa/modeling_t5_nextqa.py:544-555: mean_b || pool_b - (onehot . P)_b ||^2."""
    q = hidden_Q.mean(dim=1)
    lq = (q - ques_labels.matmul(Q_prototype).detach()).pow(2).sum(dim=1).mean()
    v = hidden_V.mean(dim=1)
    lv = (v - cate_labels.matmul(V_prototype).detach()).pow(2).sum(dim=1).mean()
    return lq, lv


# --------------------------------------------------------------------------------------
# model forward  (modeling_t5_our.py:514-713) and the train_step reduction (vqa_model.py:46-54)
# --------------------------------------------------------------------------------------
def vlt5_forward(P: Dict[str, Tensor], state: PrototypeState, cfg: Cfg, *, input_ids: Tensor,
                 vis_feats: Tensor, boxes: Tensor, labels: Optional[Tensor] = None,
                 decoder_input_ids: Optional[Tensor] = None,
                 cate_labels: Optional[Tensor] = None, ques_labels: Optional[Tensor] = None,
                 proto_update: bool = False, current_task_id: int = 0, proto_alpha: float = 0.5,
                 proto_beta: float = 0.3, memory: bool = False, training: bool = True) -> Dict[str, Tensor]:
    hidden, mask = encoder_forward(P, input_ids, vis_feats, boxes, cfg, training)
    Ls = cfg.split_L
    hq, hv = hidden[:, :Ls, :], hidden[:, Ls:, :]
    out: Dict[str, Tensor] = {}
    loss_mem_Q = loss_mem_V = 0
    if proto_update:
        cur_Q, num_Q = calculate_current_prototype(hq, ques_labels)
        cur_V, num_V = calculate_current_prototype(hv, cate_labels)
        if memory:
            loss_mem_Q, loss_mem_V = memory_loss(hq, hv, ques_labels, cate_labels,
                                                 state.Q_prototype, state.V_prototype)
        state.update(cur_Q, cur_V, num_Q.detach(), num_V.detach(), current_task_id, proto_alpha, proto_beta)
    rq, idx_Q = cosine_retrieve(state.Q_prototype, hq.mean(dim=1).detach())
    rv, idx_V = cosine_retrieve(state.V_prototype, hv.mean(dim=1).detach())
    enc_ext = torch.cat([hidden, rq.detach().unsqueeze(1), rv.detach().unsqueeze(1)], dim=1)
    B = input_ids.shape[0]
    # the decoder mask covers L + (everything after the text), i.e. the 2 prototype tokens too
    enc_mask = torch.cat([mask[:, :input_ids.shape[1]],
                          mask.new_ones(B, enc_ext.shape[1] - input_ids.shape[1])], dim=1)
    if decoder_input_ids is None:
        decoder_input_ids = shift_right(labels, cfg)
    dec_out = decoder_forward(P, decoder_input_ids, enc_ext, enc_mask, cfg, training)
    seq = dec_out * (cfg.d_model ** -0.5)                          # rescale before the tied lm_head (:661-666)
    logits = F.linear(seq, P["shared.weight"])                     # lm_head tied to shared
    # `decoder_last_hidden_state=decoder_outputs.last_hidden_state` (:699): the stack's output BEFORE the rescale
    out.update(logits=logits, encoder_hidden_states=hidden, encoder_attention_mask=enc_mask,
               max_idx_Q=idx_Q, max_idx_V=idx_V, loss_memory_Q=loss_mem_Q, loss_memory_V=loss_mem_V,
               decoder_last_hidden_state=dec_out)
    if labels is not None:
        out["loss"] = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1),
                                      ignore_index=-100, reduction="none")
    return out


def train_step_loss(loss_tok: Tensor, labels: Tensor, scores: Tensor) -> Tensor:
    """vqa_model.py:46-54: masked per-sample mean, times the answer score, batch mean."""
    m = (labels != -100).float()
    B, T = labels.shape
    l = loss_tok.view(B, T) * m
    l = l.sum(dim=1) / m.sum(dim=1).clamp(min=1)
    return (l * scores).mean()


# --------------------------------------------------------------------------------------
# parameters, init, optimizer
# --------------------------------------------------------------------------------------
def param_shapes(cfg: Cfg) -> Dict[str, Tuple[int, ...]]:
    """state_dict names and shapes of the reference model (tied aliases omitted)."""
    d, inner, ff, H = cfg.d_model, cfg.inner, cfg.d_ff, cfg.num_heads
    s: Dict[str, Tuple[int, ...]] = {"shared.weight": (cfg.vocab_size, d)}
    ve = "encoder.visual_embedding."
    s[ve + "feat_embedding.0.weight"] = (d, cfg.feat_dim)
    s[ve + "feat_embedding.0.bias"] = (d,)
    s[ve + "feat_embedding.1.weight"] = (d,)
    s[ve + "absolute_vis_pos_embedding.0.weight"] = (d, cfg.pos_dim + 1)
    s[ve + "absolute_vis_pos_embedding.0.bias"] = (d,)
    s[ve + "absolute_vis_pos_embedding.1.weight"] = (d,)
    s[ve + "img_order_embedding.weight"] = (cfg.n_images, d)

    def attn(prefix):
        s[prefix + "q.weight"] = (inner, d)
        s[prefix + "k.weight"] = (inner, d)
        s[prefix + "v.weight"] = (inner, d)
        s[prefix + "o.weight"] = (d, inner)

    def ffn(prefix):
        if cfg.gated_act:
            s[prefix + "wi_0.weight"] = (ff, d)
            s[prefix + "wi_1.weight"] = (ff, d)
        else:
            s[prefix + "wi.weight"] = (ff, d)
        s[prefix + "wo.weight"] = (d, ff)

    for i in range(cfg.num_layers):
        pre = f"encoder.block.{i}.layer."
        attn(pre + "0.SelfAttention.")
        if i == 0:
            s[pre + "0.SelfAttention.relative_attention_bias.weight"] = (cfg.rel_buckets, H)
        s[pre + "0.layer_norm.weight"] = (d,)
        ffn(pre + "1.DenseReluDense.")
        s[pre + "1.layer_norm.weight"] = (d,)
    s["encoder.final_layer_norm.weight"] = (d,)
    for i in range(cfg.num_decoder_layers):
        pre = f"decoder.block.{i}.layer."
        attn(pre + "0.SelfAttention.")
        if i == 0:
            s[pre + "0.SelfAttention.relative_attention_bias.weight"] = (cfg.rel_buckets, H)
        s[pre + "0.layer_norm.weight"] = (d,)
        attn(pre + "1.EncDecAttention.")
        s[pre + "1.layer_norm.weight"] = (d,)
        ffn(pre + "2.DenseReluDense.")
        s[pre + "2.layer_norm.weight"] = (d,)
    s["decoder.final_layer_norm.weight"] = (d,)
    s["prototype_fc1.weight"] = (d, d)
    s["prototype_fc1.bias"] = (d,)
    s["prototype_fc2.weight"] = (d, d)
    s["prototype_fc2.bias"] = (d,)
    return s


def init_params(cfg: Cfg, seed: int = 0) -> Dict[str, Tensor]:
    """Random init with the reference's distributions (trainer_base.py:218-238 then the T5
    re-init): everything N(0,1) first, then q ~ N(0,(d*d_kv)^-1/2), k,v,wi ~ N(0,d^-1/2),
    o ~ N(0,(H*d_kv)^-1/2), wo ~ N(0,d_ff^-1/2), rel-bias ~ N(0,d^-1/2), norms = 1, biases = 0.
    The draw ORDER is ours (the reference model cannot be built here), so a seed does not
    reproduce the reference's tensors; fixtures therefore ship explicit weights."""
    g = torch.Generator().manual_seed(seed)
    d = cfg.d_model
    P: Dict[str, Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("layer_norm.weight") or name.endswith("embedding.1.weight"):
            P[name] = torch.ones(shape)
        elif name.endswith(".bias"):
            P[name] = torch.zeros(shape)
        else:
            std = 1.0
            if "Attention.q." in name:
                std = (d * cfg.d_kv) ** -0.5
            elif "Attention.k." in name or "Attention.v." in name or "relative_attention_bias" in name:
                std = d ** -0.5
            elif "Attention.o." in name:
                std = cfg.inner ** -0.5
            elif ".wi" in name:
                std = d ** -0.5
            elif ".wo." in name:
                std = cfg.d_ff ** -0.5
            P[name] = torch.randn(shape, generator=g) * std
    return P


def weight_decay_of(name: str, wd: float) -> float:
    """trainer_base.py:148-161: substring match on "bias" / "LayerNorm.weight"; T5 norms are
    called `layer_norm.weight`, so only real biases are exempt (and the rel-pos "..._bias.weight")."""
    return 0.0 if ("bias" in name or "LayerNorm.weight" in name) else wd


class HFAdamW:
    """`transformers.optimization.AdamW` of 4.2.1 (bias-corrected, eps outside the correction,
    decoupled decay applied after the Adam update), lr = 1e-4, eps = 1e-6, wd = 0.01."""

    def __init__(self, named: Dict[str, Tensor], lr=1e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01):
        self.named, self.lr, self.betas, self.eps, self.wd = named, lr, betas, eps, weight_decay
        self.m = {k: torch.zeros_like(v) for k, v in named.items()}
        self.v = {k: torch.zeros_like(v) for k, v in named.items()}
        self.t = {k: 0 for k in named}

    @torch.no_grad()
    def step(self, lr_scale: float = 1.0) -> None:
        b1, b2 = self.betas
        lr = self.lr * lr_scale
        for k, p in self.named.items():
            if p.grad is None:
                continue
            g = p.grad
            self.t[k] += 1
            t = self.t[k]
            self.m[k].mul_(b1).add_(g, alpha=1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = self.v[k].sqrt().add_(self.eps)
            step_size = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
            p.addcdiv_(self.m[k], denom, value=-step_size)
            wd = weight_decay_of(k, self.wd)
            if wd > 0:
                p.add_(p, alpha=-lr * wd)


def clip_grad_norm(params, max_norm: float) -> Tensor:
    """torch.nn.utils.clip_grad_norm_ semantics: coef = max_norm / (total + 1e-6), clamped to 1."""
    grads = [p.grad for p in params if p.grad is not None]
    total = torch.sqrt(sum((g.float() ** 2).sum() for g in grads))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def warmup_constant_lr_scale(step: int, warmup_iters: int) -> float:
    """get_constant_schedule_with_warmup: step / max(1, warmup) until warmup, then 1."""
    return float(step) / float(max(1.0, warmup_iters)) if step < warmup_iters else 1.0


# --------------------------------------------------------------------------------------
# a small stateful wrapper used by tests and bench.py's cpu_baseline leg
# --------------------------------------------------------------------------------------
class OracleModel:
    def __init__(self, cfg: Cfg, params: Optional[Dict[str, Tensor]] = None, seed: int = 0):
        self.cfg = cfg
        self.P = {k: v.clone().requires_grad_(True) for k, v in (params or init_params(cfg, seed)).items()}
        self.state = PrototypeState()
        self.used = {k: v for k, v in self.P.items() if not k.startswith("prototype_fc")}

    def train_step(self, batch: Dict[str, Tensor], current_task_id: int, proto_alpha: float,
                   proto_beta: float, training: bool = True) -> Dict[str, Tensor]:
        out = vlt5_forward(self.P, self.state, self.cfg, input_ids=batch["input_ids"],
                           vis_feats=batch["vis_feats"], boxes=batch["boxes"], labels=batch["target_ids"],
                           cate_labels=batch["cate_labels"], ques_labels=batch["ques_labels"],
                           proto_update=True, current_task_id=current_task_id, proto_alpha=proto_alpha,
                           proto_beta=proto_beta, training=training)
        out["loss_tok"] = out["loss"]
        out["loss"] = train_step_loss(out["loss_tok"], batch["target_ids"], batch["scores"])
        return out

    def zero_grad(self):
        for p in self.P.values():
            p.grad = None


def synthetic_batch(cfg: Cfg, B: int, L: int = 20, V: int = 36, T: int = 5, seed: int = 66666,
                    task_id: int = 0, cate_group: int = 0) -> Dict[str, Tensor]:
    """Seeded synthetic batch of SURVEY 8d: non-negative sparse-ish region features, sorted
    box corners in [0,1], ragged questions (pad 0, one row forced to full length), ragged
    answers ending in EOS (pad -100), one-hot task / category labels, answer scores."""
    g = torch.Generator().manual_seed(seed)
    feats = torch.relu(torch.randn(B, V, cfg.feat_dim, generator=g)) * 1.5
    xs = torch.rand(B, V, 2, generator=g).sort(dim=2).values
    ys = torch.rand(B, V, 2, generator=g).sort(dim=2).values
    boxes = torch.stack([xs[..., 0], ys[..., 0], xs[..., 1], ys[..., 1]], dim=2)
    hi = min(32000, cfg.vocab_size - 100)
    ids = torch.randint(2, hi, (B, L), generator=g)
    lens = torch.randint(min(6, L), L + 1, (B,), generator=g)
    lens[0] = L
    ids = ids * (torch.arange(L)[None, :] < lens[:, None])
    tgt = torch.randint(2, hi, (B, T), generator=g)
    tl = torch.randint(2, T + 1, (B,), generator=g) if T > 2 else torch.full((B,), T)
    tl[0] = T
    pos = torch.arange(T)[None, :]
    tgt = torch.where(pos == (tl[:, None] - 1), torch.ones_like(tgt), tgt)     # EOS = 1
    tgt = torch.where(pos < tl[:, None], tgt, torch.full_like(tgt, -100))
    ques = torch.zeros(B, cfg.n_ques)
    ques[:, task_id] = 1
    cate_ids = cate_group * 16 + torch.randint(0, 16, (B,), generator=g)
    cate = torch.zeros(B, cfg.n_cate).scatter_(1, cate_ids[:, None] % cfg.n_cate, 1.0)
    choice = torch.tensor([0.3, 0.6, 0.9, 1.0])
    scores = choice[torch.randint(0, 4, (B,), generator=g)]
    return dict(vis_feats=feats, boxes=boxes, input_ids=ids, target_ids=tgt, cate_labels=cate,
                ques_labels=ques, scores=scores)
