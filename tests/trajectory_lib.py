"""TEST INFRASTRUCTURE -- accuracy proxy: the engine and the fp32 oracle trained side by side on the same continual schedule.

The reference's headline claim is final-task VQA accuracy; its datasets and partitions are not available offline, so the stand-in is a
synthetic VQA problem with a LEARNABLE rule (the answer is a fixed function of question type, an attribute token and the image's
cluster) trained through the dual-level loop of `Trainer.train` (VL-T5/src/vqacl.py:314-373): outer loop over question-type tasks,
inner loop over category groups, a NEW optimizer + constant-with-warm-up schedule per (task, group) (trainer_base.py:130-198), and --
from the second task on -- every current batch followed by a rehearsal batch of earlier tasks (vqacl.py:355-369), clip 5 + AdamW.

Both sides start from the same weights and see the same batches.  The checker is `oracle/ref_cpu.py` run with torch on the SAME GPU
(fp32, eager: ~1.2 k samples/s), so a 300-step trajectory at B = 80 takes seconds, not hours.  Reported:
  * dropout off: |loss_engine - loss_oracle| per optimizer step, prototype-index agreement per step, and after training the greedy
    answers of the two trained models on held-out questions: agreement with each other and accuracy of each against the rule;
  * dropout 0.1 (different RNGs by construction): several seeds per side, mean loss curves compared in windows (z-scores), held-out
    accuracy per seed.
Used by tests/test_gpu_trajectory.py (short) and tools/trajectory.py (the full figures committed under profiles/).
"""
import math
import time

import torch

EOS, PAD = 1, 0


class SyntheticVQA:
    """A VQA-shaped problem with a rule to learn.  An image belongs to one of `clusters` clusters of its category group (all 36 regions
    = relu(centre + noise)); a question is [task token, attribute token, filler ...] padded to L; the answer (1-3 tokens + EOS) is a fixed
    table entry of (task, attribute, group, cluster).  Everything is generated on `device` from a seeded generator, so the engine and
    the oracle are handed the very same tensors."""

    def __init__(self, device, seed=1234, n_tasks=3, n_groups=2, clusters=8, n_attr=4, L=20, V=36, T=5, feat_dim=2048, n_ques=10, n_cate=80,
                 answer_vocab=48, noise=0.5):
        self.dev, self.L, self.V, self.T, self.fd = device, L, V, T, feat_dim
        self.n_tasks, self.n_groups, self.clusters, self.n_attr = n_tasks, n_groups, clusters, n_attr
        self.n_ques, self.n_cate, self.noise = n_ques, n_cate, noise
        g = torch.Generator().manual_seed(seed)
        self.centres = (torch.randn(n_groups, clusters, feat_dim, generator=g)).to(device)
        n_ans = torch.randint(1, 4, (n_tasks, n_attr, n_groups, clusters), generator=g)
        toks = 3000 + torch.randint(0, answer_vocab, (n_tasks, n_attr, n_groups, clusters, 3), generator=g)
        ans = torch.full((n_tasks, n_attr, n_groups, clusters, T), -100, dtype=torch.long)
        for idx in torch.cartesian_prod(*[torch.arange(n) for n in (n_tasks, n_attr, n_groups, clusters)]).tolist():
            n = int(n_ans[tuple(idx)])
            ans[tuple(idx)][:n] = toks[tuple(idx)][:n]
            ans[tuple(idx)][n] = EOS
        self.answers = ans.to(device)

    def batch(self, B, task, group, seed, tasks_from=None):
        """A batch of question-type `task` (or, `tasks_from` = list of earlier tasks, a rehearsal batch drawn from those) in category
        group `group`.  Returns the collate_fn schema + 'answer' (the rule's tokens, for scoring)."""
        dev = self.dev
        g = torch.Generator(device=dev).manual_seed(int(seed))
        if tasks_from:
            t = torch.tensor(tasks_from, device=dev)[torch.randint(0, len(tasks_from), (B,), device=dev, generator=g)]
        else:
            t = torch.full((B,), task, dtype=torch.long, device=dev)
        a = torch.randint(0, self.n_attr, (B,), device=dev, generator=g)
        k = torch.randint(0, self.clusters, (B,), device=dev, generator=g)
        c = self.centres[group][k]                                                      # [B, fd]
        feats = torch.relu(c[:, None, :] + self.noise * torch.randn(B, self.V, self.fd, device=dev, generator=g)) * 1.5
        xs = torch.rand(B, self.V, 2, device=dev, generator=g).sort(dim=2).values
        ys = torch.rand(B, self.V, 2, device=dev, generator=g).sort(dim=2).values
        boxes = torch.stack([xs[..., 0], ys[..., 0], xs[..., 1], ys[..., 1]], dim=2)
        ids = torch.randint(1000, 1100, (B, self.L), device=dev, generator=g)
        lens = torch.randint(6, self.L + 1, (B,), device=dev, generator=g)
        lens[0] = self.L
        ids = ids * (torch.arange(self.L, device=dev)[None, :] < lens[:, None])
        ids[:, 0] = 100 + t
        ids[:, 1] = 200 + a
        tgt = self.answers[t, a, group, k]                                              # [B, T]
        ques = torch.zeros(B, self.n_ques, device=dev).scatter_(1, t[:, None], 1.0)
        cate = torch.zeros(B, self.n_cate, device=dev).scatter_(1, (group * 16 + k)[:, None] % self.n_cate, 1.0)
        scores = torch.tensor([0.3, 0.6, 0.9, 1.0], device=dev)[torch.randint(0, 4, (B,), device=dev, generator=g)]
        return dict(vis_feats=feats, boxes=boxes, input_ids=ids, target_ids=tgt, ques_labels=ques, cate_labels=cate, scores=scores,
                    answer=tgt)


def schedule(n_tasks, n_groups, steps_per_stage):
    """[(task, group, step_in_stage, steps_in_stage, rehearsal)] in the order Trainer.train runs them: task 0 has no memory; later tasks
    alternate current / rehearsal batches (each its own optimizer step)."""
    out = []
    for task in range(n_tasks):
        for group in range(n_groups):
            for i in range(steps_per_stage):
                out.append((task, group, i, steps_per_stage, task > 0 and (i % 2 == 1)))
    return out


def warmup_scale(step, warmup_iters):
    """get_constant_schedule_with_warmup as LambdaLR applies it: the k-th optimizer step of a stage runs with lambda(k)."""
    return float(step) / float(max(1, warmup_iters)) if step < warmup_iters else 1.0


def answers_from_tokens(tok):
    """Greedy token rows [B, n] (column 0 = start token) -> tuples of the tokens before the first EOS."""
    out = []
    for row in tok.tolist():
        seq = []
        for v in row[1:]:
            if v == EOS:
                break
            seq.append(v)
        out.append(tuple(seq))
    return out


def rule_answers(tgt):
    return [tuple(v for v in row if v not in (EOS, -100)) for row in tgt.tolist()]


class EngineSide:
    def __init__(self, dev, ocfg, params, dropout, seed, lr):
        from vqacl_amd import VLT5VQA, VLT5Config
        self.cfg = VLT5Config(d_model=ocfg.d_model, d_kv=ocfg.d_kv, num_heads=ocfg.num_heads, d_ff=ocfg.d_ff, num_layers=ocfg.num_layers,
                              num_decoder_layers=ocfg.num_decoder_layers, vocab_size=ocfg.vocab_size, feat_dim=ocfg.feat_dim,
                              dropout_rate=dropout, n_ques=ocfg.n_ques, n_cate=ocfg.n_cate)
        self.model = VLT5VQA(self.cfg, device=dev)
        self.model.load_state_dict({k: v.detach() for k, v in params.items()}, strict=False)
        self.model.base_seed = 0x5EED + 7919 * seed
        self.lr, self.opt = lr, None

    def new_stage(self):
        from vqacl_amd import FusedAdamW, reference_param_groups
        self.opt = FusedAdamW(reference_param_groups(self.model, 0.01), self.model, lr=self.lr, eps=1e-6, max_grad_norm=5.0)

    def step(self, batch, task, scale):
        self.model.train()
        for g in self.opt.param_groups:
            g["lr"] = self.lr * scale
        res = self.model.train_step(batch, task, 0.5, 0.3)
        res["loss"].backward()
        self.opt.step()
        for p in self.model.parameters():
            p.grad = None
        return res["loss"].detach(), self.model._cached_idx

    @torch.no_grad()
    def greedy(self, batch, max_length):
        self.model.eval()
        return self.model.greedy_generate(batch["input_ids"], (batch["vis_feats"], batch["boxes"]), max_length=max_length, eos_token_id=EOS)


class OracleSide:
    """oracle/ref_cpu.py with its tensors on the GPU (plain torch eager ops, fp32)."""

    def __init__(self, dev, ocfg, params, dropout, seed, lr):
        from oracle import ref_cpu as R
        self.R, self.dev, self.lr = R, dev, lr
        import dataclasses
        self.cfg = dataclasses.replace(ocfg, dropout=dropout)
        self.model = R.OracleModel(self.cfg, {k: v.to(dev) for k, v in params.items()})
        self.gen_seed = 4242 + seed
        self.opt = None
        torch.manual_seed(self.gen_seed)
        torch.cuda.manual_seed(self.gen_seed)

    def new_stage(self):
        self.opt = self.R.HFAdamW(self.model.used, lr=self.lr, eps=1e-6, weight_decay=0.01)

    def step(self, batch, task, scale):
        R = self.R
        with torch.device(self.dev):         # the restatement builds its index tensors with default-device factories
            self.model.zero_grad()
            out = self.model.train_step(batch, task, 0.5, 0.3, training=True)
            out["loss"].backward()
            R.clip_grad_norm(list(self.model.used.values()), 5.0)
            self.opt.step(lr_scale=scale)
        return out["loss"].detach(), (out["max_idx_Q"], out["max_idx_V"])

    @torch.no_grad()
    def greedy(self, batch, max_length):
        R, B = self.R, batch["input_ids"].shape[0]
        cur = torch.zeros(B, 1, dtype=torch.long, device=self.dev)
        done = torch.zeros(B, dtype=torch.bool, device=self.dev)
        for _ in range(max_length - 1):
            with torch.device(self.dev):
                o = R.vlt5_forward(self.model.P, self.model.state, self.cfg, input_ids=batch["input_ids"], vis_feats=batch["vis_feats"],
                                   boxes=batch["boxes"], decoder_input_ids=cur, training=False)
            nxt = o["logits"][:, -1].argmax(-1)
            nxt = torch.where(done, torch.full_like(nxt, PAD), nxt)
            cur = torch.cat([cur, nxt[:, None]], dim=1)
            done = done | (nxt == EOS)
        return cur


def heldout_eval(data, sides, n_eval, B, seed0=900000):
    """Greedy answers of every side on the same held-out questions (all tasks and groups, fresh noise).  Returns per side the list of
    answers and the rule's answers."""
    got = [[] for _ in sides]
    truth = []
    n_stage = data.n_tasks * data.n_groups
    per = max(1, n_eval // n_stage)
    i = 0
    for task in range(data.n_tasks):
        for group in range(data.n_groups):
            left = per
            while left > 0:
                b = min(B, left)
                batch = data.batch(b, task, group, seed0 + i)
                i += 1
                truth += rule_answers(batch["answer"])
                for s, side in enumerate(sides):
                    got[s] += answers_from_tokens(side.greedy(batch, data.T + 1))
                left -= b
    return got, truth


def run_pair(dev, ocfg, dropout=0.0, seed=0, B=80, steps_per_stage=50, n_tasks=3, n_groups=2, lr=1e-4, n_eval=512, data_seed=1234,
             init_seed=3, sides=("engine", "oracle"), log=None):
    """Train the requested sides in lockstep on the same batches.  Returns losses [n_sides][steps], index agreement per step (both
    sides present), held-out answers and wall time per side."""
    from oracle import ref_cpu as R
    data = SyntheticVQA(dev, seed=data_seed, n_tasks=n_tasks, n_groups=n_groups, feat_dim=ocfg.feat_dim, n_ques=ocfg.n_ques, n_cate=ocfg.n_cate)
    params = R.init_params(ocfg, seed=init_seed)
    if True:
        objs = []
        for name in sides:
            objs.append((EngineSide if name == "engine" else OracleSide)(dev, ocfg, params, dropout, seed, lr))
        plan = schedule(n_tasks, n_groups, steps_per_stage)
        losses = [[] for _ in objs]
        agree_q, agree_v = [], []
        wall = [0.0 for _ in objs]
        for n, (task, group, i, n_stage, rehearsal) in enumerate(plan):
            if i == 0:
                for o in objs:
                    o.new_stage()
            scale = warmup_scale(i, int(n_stage * 0.05))
            batch = data.batch(B, task, group, 10_000 + n, tasks_from=list(range(task)) if rehearsal else None)
            idx = []
            for s, o in enumerate(objs):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                loss, ix = o.step(batch, task, scale)
                torch.cuda.synchronize()
                wall[s] += time.perf_counter() - t0
                losses[s].append(float(loss))
                idx.append(ix)
            if len(objs) == 2:
                agree_q.append(float((idx[0][0].to(dev) == idx[1][0].to(dev)).float().mean()))
                agree_v.append(float((idx[0][1].to(dev) == idx[1][1].to(dev)).float().mean()))
            if log and (n % 25 == 0 or n == len(plan) - 1):
                log(f"  step {n:4d} task {task} group {group} {'mem' if rehearsal else 'cur'} lr x{scale:.2f}  " +
                    "  ".join(f"{name} {losses[s][-1]:.4f}" for s, name in enumerate(sides)))
        answers, truth = heldout_eval(data, objs, n_eval, min(B, 128))
    return dict(sides=list(sides), losses=losses, agree_q=agree_q, agree_v=agree_v, answers=answers, truth=truth, wall_s=wall,
                plan=plan)


def summarize_pair(r):
    """Figures of a dropout-off engine / oracle pair."""
    le, lo = r["losses"]
    d = [abs(a - b) for a, b in zip(le, lo)]
    n = len(d)
    ae, ao = r["answers"]
    truth = r["truth"]
    acc_e = sum(a == t for a, t in zip(ae, truth)) / len(truth)
    acc_o = sum(a == t for a, t in zip(ao, truth)) / len(truth)
    agree = sum(a == b for a, b in zip(ae, ao)) / len(truth)
    return dict(steps=n, max_dloss=max(d), mean_dloss=sum(d) / n, max_dloss_first50=max(d[:50]), dloss_last50_mean=sum(d[-50:]) / len(d[-50:]),
                loss_first=(le[0], lo[0]), loss_last10=(sum(le[-10:]) / 10, sum(lo[-10:]) / 10),
                idx_agree_q=sum(r["agree_q"]) / n, idx_agree_v=sum(r["agree_v"]) / n,
                heldout=len(truth), acc_engine=acc_e, acc_oracle=acc_o, answer_agreement=agree,
                wall_engine_s=r["wall_s"][0], wall_oracle_s=r["wall_s"][1])


def compare_seeds(curves_e, curves_o, window=10):
    """Mean loss curves of several seeds per side, compared window by window: z = (mean_e - mean_o) / sqrt(var_e / n_e + var_o / n_o)
    with the per-seed window means as samples.  Returns the z-scores and the share of windows with |z| <= 2."""
    n = min(len(c) for c in curves_e + curves_o)
    zs = []
    for a in range(0, n - window + 1, window):
        we = [sum(c[a:a + window]) / window for c in curves_e]
        wo = [sum(c[a:a + window]) / window for c in curves_o]
        me, mo = sum(we) / len(we), sum(wo) / len(wo)
        ve = sum((x - me) ** 2 for x in we) / max(1, len(we) - 1)
        vo = sum((x - mo) ** 2 for x in wo) / max(1, len(wo) - 1)
        se = math.sqrt(ve / len(we) + vo / len(wo))
        zs.append((a, me, mo, (me - mo) / se if se > 0 else 0.0, se))
    within = sum(abs(z[3]) <= 2.0 for z in zs) / len(zs)
    return zs, within


class _Moments:
    """Running first / second moments over dropout seeds (f64 on the device), with the first moment also kept per half (even / odd
    seeds) so the seed noise left in a mean can be read off the data: cos(mean_A, mean_B)."""

    def __init__(self):
        self.n, self.s, self.q, self.half = 0, {}, {}, ({}, {})

    def add(self, **xs):
        for k, x in xs.items():
            x = x.detach().double().reshape(-1)
            if k not in self.s:
                self.s[k], self.q[k] = torch.zeros_like(x), torch.zeros_like(x)
                self.half[0][k], self.half[1][k] = torch.zeros_like(x), torch.zeros_like(x)
            self.s[k] += x
            self.q[k] += x * x
            self.half[self.n & 1][k] += x
        self.n += 1

    def mean(self, k):
        return self.s[k] / self.n

    def var_sum(self, k):
        """Sum over the elements of the unbiased per-element variance over the seeds."""
        m = self.mean(k)
        return float(((self.q[k] - self.n * m * m) / (self.n - 1)).clamp(min=0).sum())


def _cos(a, b):
    return float(torch.dot(a, b) / (a.norm() * b.norm()).clamp(min=1e-300))


def dropout_moments(dev, ocfg, B=80, K=32, p=0.1, warm_steps=0, init_seed=3, data_seed=1234, lr=1e-4, log=None):
    """The benched configuration (dropout on) against the oracle WITHOUT a shared random stream: from the same weights and on the same
    batch, K dropout seeds per side (the engine's counter-hash masks, torch's generator); what must agree is the DISTRIBUTION over the
    seeds -- the mean and the per-element variance of the loss, the encoder output, the decoder output, the logits and the gradient.
    A dropout site that is missing, doubled, applied with another probability or scale, or whose backward uses another mask than its
    forward shows up as a variance ratio off 1 or as a mean gradient that points elsewhere (oracle alone, one of the 24 sites of the
    tiny model removed: variance ratios 0.57-0.99, most sites 0.93-0.97).

    warm_steps > 0: that many dropout-off optimizer steps of the ENGINE first, both sides then start from its weights (away from the
    initialisation, where the loss barely reacts to the masks).  The prototype memory is initialised by one eval-mode train_step per
    side and then left alone (proto_update=False), so the seeds are independent draws."""
    from oracle import ref_cpu as R
    import dataclasses
    data = SyntheticVQA(dev, seed=data_seed, feat_dim=ocfg.feat_dim, n_ques=ocfg.n_ques, n_cate=ocfg.n_cate)
    params = R.init_params(ocfg, seed=init_seed)
    eng = EngineSide(dev, ocfg, params, 0.0, 0, lr)
    if warm_steps:
        eng.new_stage()
        for i in range(warm_steps):
            eng.step(data.batch(B, 0, 0, 20_000 + i), 0, warmup_scale(i, max(1, warm_steps // 20)))
        sd = eng.model.state_dict()
        params = {k: sd[k].detach().float().cpu().clone() for k in params}
    batch = data.batch(B, 0, 0, 77_777)
    names = [k for k in params if not k.startswith("prototype_fc")]
    ocfg_p = dataclasses.replace(ocfg, dropout=p)
    orc = R.OracleModel(ocfg_p, {k: v.to(dev) for k, v in params.items()})
    model = eng.model
    # prototype memory: one eval-mode step per side
    model.eval()
    with torch.no_grad():
        model.train_step(batch, 0, 0.5, 0.3)
        with torch.device(dev):
            orc.train_step(batch, 0, 0.5, 0.3, training=False)
    fwd_kw = dict(input_ids=batch["input_ids"], labels=batch["target_ids"], cate_labels=batch["cate_labels"], ques_labels=batch["ques_labels"],
                  proto_update=False)

    def engine_draw(seed, training=True):
        model.train(training)
        model.cfg.dropout_rate = p
        model.base_seed, model._step_count = (0x5EED + 7919 * seed) & 0x7FFFFFFF, 0
        for q in model.parameters():
            q.grad = None
        out = model(vis_inputs=(batch["vis_feats"], batch["boxes"]), scores=batch["scores"], **fwd_kw)
        out["loss_reduced"].backward()
        g = torch.cat([model._params_by_name[k].grad.reshape(-1) for k in names])
        return dict(loss=out["loss_reduced"].reshape(1), enc=out["encoder_hidden_states"], dec=out["decoder_last_hidden_state"],
                    logits=out["logits"], grad=g)

    def oracle_draw(seed, training=True):
        torch.manual_seed(1_000_003 * (seed + 1))
        torch.cuda.manual_seed(1_000_003 * (seed + 1))
        orc.zero_grad()
        with torch.device(dev):
            out = R.vlt5_forward(orc.P, orc.state, ocfg_p, vis_feats=batch["vis_feats"], boxes=batch["boxes"], training=training, **fwd_kw)
            loss = R.train_step_loss(out["loss"], batch["target_ids"], batch["scores"])
            loss.backward()
        g = torch.cat([orc.P[k].grad.reshape(-1) for k in names])
        return dict(loss=loss.reshape(1), enc=out["encoder_hidden_states"], dec=out["decoder_last_hidden_state"], logits=out["logits"], grad=g)

    # the masks off: the scale of the bf16 difference the means can at best agree to
    e0, o0 = engine_draw(0, training=False), oracle_draw(0, training=False)
    e0, o0 = {k: v.detach().double().reshape(-1) for k, v in e0.items()}, {k: v.detach().double().reshape(-1) for k, v in o0.items()}
    base = {k: float((e0[k] - o0[k]).norm() / o0[k].norm().clamp(min=1e-300)) for k in e0}
    base["grad_cos"] = _cos(e0["grad"], o0["grad"])
    me, mo = _Moments(), _Moments()
    for s in range(K):
        me.add(**engine_draw(s))
        mo.add(**oracle_draw(s))
        if log and (s + 1) % 8 == 0:
            log(f"  {s + 1} seeds per side")
    r = dict(K=K, B=B, p=p, warm_steps=warm_steps, masks_off=base, keys={})
    le, lo = me.mean("loss").item(), mo.mean("loss").item()
    se = math.sqrt((me.var_sum("loss") + mo.var_sum("loss")) / K)
    r["loss"] = dict(engine=le, oracle=lo, se=se, z=(le - lo) / se if se > 0 else 0.0, sd_engine=math.sqrt(me.var_sum("loss")),
                     sd_oracle=math.sqrt(mo.var_sum("loss")), masks_off=(float(e0["loss"]), float(o0["loss"])))
    for k in ("enc", "dec", "logits", "grad"):
        a, b = me.mean(k), mo.mean(k)
        ve, vo = me.var_sum(k), mo.var_sum(k)
        r["keys"][k] = dict(var_ratio=ve / vo if vo > 0 else float("nan"),
                            mean_rel_diff=float((a - b).norm() / b.norm()), noise_floor=math.sqrt((ve + vo) / K) / float(b.norm()),
                            cos_means=_cos(a, b), cos_halves_engine=_cos(me.half[0][k], me.half[1][k]),
                            cos_halves_oracle=_cos(mo.half[0][k], mo.half[1][k]), norm_ratio=float(a.norm() / b.norm()),
                            noise_to_mean=math.sqrt(vo) / float(b.norm()))
    return r


def format_moments(r):
    out = [f"dropout {r['p']}: {r['K']} seeds per side, B = {r['B']}, same weights ({r['warm_steps']} dropout-off optimizer steps from the initialisation) and batch",
           f"  loss            engine {r['loss']['engine']:.5f}  oracle {r['loss']['oracle']:.5f}  difference {r['loss']['engine'] - r['loss']['oracle']:+.5f} = {r['loss']['z']:+.2f} standard errors "
           f"(sd over seeds engine {r['loss']['sd_engine']:.5f} oracle {r['loss']['sd_oracle']:.5f}; masks off {r['loss']['masks_off'][0]:.5f} / {r['loss']['masks_off'][1]:.5f})",
           "  quantity   variance over seeds engine/oracle   |mean_e - mean_o| / |mean_o|  (seed noise left in it, masks-off difference)   cos(mean_e, mean_o)  (halves: engine, oracle)   |mean_e| / |mean_o|   sd / |mean|"]
    for k, v in r["keys"].items():
        out.append(f"  {k:8s}   {v['var_ratio']:.4f}                               {v['mean_rel_diff']:.5f}  ({v['noise_floor']:.5f}, {r['masks_off'][k]:.5f})"
                   f"                                  {v['cos_means']:.5f}  ({v['cos_halves_engine']:.5f}, {v['cos_halves_oracle']:.5f})          {v['norm_ratio']:.4f}   {v['noise_to_mean']:.3f}")
    out.append(f"  masks off: cos(grad_e, grad_o) {r['masks_off']['grad_cos']:.5f}")
    return out


def first_stage_seeds(dev, ocfg, seeds=32, steps=60, B=80, lr=1e-4, lo=10, log=None):
    """Many dropout seeds per side over the FIRST stage only (same weights at step 0, same batches): the per-seed mean loss over steps
    lo..steps as the sample; returns means, standard deviations and z of the difference.  (The sharper look at 'does one side learn
    faster under dropout' than the windows of the full schedule, whose neighbouring windows are not independent.)"""
    stat = ([], [])
    curves = ([], [])
    for seed in range(seeds):
        for s, name in enumerate(("engine", "oracle")):
            r = run_pair(dev, ocfg, dropout=0.1, seed=seed, B=B, steps_per_stage=steps, n_tasks=1, n_groups=1, lr=lr, n_eval=1, sides=(name,))
            c = r["losses"][0]
            curves[s].append(c)
            stat[s].append(sum(c[lo:]) / len(c[lo:]))
        if log and (seed + 1) % 4 == 0:
            log(f"  {seed + 1} seeds per side")

    def ms(v):
        m = sum(v) / len(v)
        return m, (sum((x - m) ** 2 for x in v) / (len(v) - 1)) ** 0.5
    (me, sde), (mo, sdo) = ms(stat[0]), ms(stat[1])
    se = math.sqrt(sde ** 2 / seeds + sdo ** 2 / seeds)
    zs, within = compare_seeds(curves[0], curves[1], window=10)
    return dict(seeds=seeds, steps=steps, lo=lo, mean_engine=me, mean_oracle=mo, sd_engine=sde, sd_oracle=sdo, z=(me - mo) / se, se=se,
                windows=zs, per_seed=stat)
