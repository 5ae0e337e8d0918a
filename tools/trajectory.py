#!/usr/bin/env python3
"""Accuracy proxy: the engine and the fp32 oracle (run with torch on the same GPU) trained side by side on a
synthetic VQA problem with a learnable rule, through the dual-level continual schedule of Trainer.train (vqacl.py:314-373).

    python tools/trajectory.py --out profiles/r05_trajectory.txt          # VL-T5-base, B = 80, 3 tasks x 2 groups x 50 steps = 300

The logic lives in tests/trajectory_lib.py (test infrastructure: it imports the oracle); this is the command line around it.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=80)
    ap.add_argument("--steps-per-stage", type=int, default=50)
    ap.add_argument("--tasks", type=int, default=3)
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--eval", type=int, default=512)
    ap.add_argument("--seeds", type=int, default=5, help="seeds per side of the dropout-0.1 comparison (0: skip it)")
    ap.add_argument("--first-stage-seeds", type=int, default=0, help="only this: N dropout seeds per side over the first stage (steps-per-stage steps)")
    ap.add_argument("--tiny", action="store_true", help="the tiny configuration (a quick look, not the figure)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    from oracle import ref_cpu as R
    from tests import trajectory_lib as T
    assert torch.cuda.is_available(), "needs the GPU"
    dev = torch.device("cuda", 0)
    ocfg = R.tiny_cfg(vocab_size=3200, feat_dim=256) if args.tiny else R.Cfg(dropout=0.0)
    lines = []

    def log(x=""):
        print(x, flush=True)
        lines.append(x)
    from vqacl_amd.build import source_hash
    log(f"# trajectory proxy: {'tiny' if args.tiny else 'VL-T5-base'}, B = {args.batch}, {args.tasks} tasks x {args.groups} category groups x "
        f"{args.steps_per_stage} optimizer steps = {args.tasks * args.groups * args.steps_per_stage} steps; new AdamW + 5 % warm-up per (task, group), "
        f"clip 5, lr {args.lr:g}; from task 1 on every second step is a rehearsal batch of earlier tasks")
    log(f"# source_sha16 {source_hash()}   {time.strftime('%Y-%m-%d %H:%M:%S')}   checker: oracle/ref_cpu.py as torch eager fp32 on the same GPU")
    if args.first_stage_seeds:
        log(f"\n## dropout 0.1, first stage only ({args.steps_per_stage} steps from the same weights on the same batches), {args.first_stage_seeds} seeds per side")
        r = T.first_stage_seeds(dev, ocfg, seeds=args.first_stage_seeds, steps=args.steps_per_stage, B=args.batch, lr=args.lr, log=log)
        log(f"per-seed mean loss over steps {r['lo']}..{r['steps']}: engine {r['mean_engine']:.5f} (sd {r['sd_engine']:.5f})  oracle {r['mean_oracle']:.5f} (sd {r['sd_oracle']:.5f})  "
            f"difference {r['mean_engine'] - r['mean_oracle']:+.5f} = {r['z']:+.2f} standard errors ({r['se']:.5f})")
        log("  window start, mean engine, mean oracle, z, sigma")
        for a, me, mo, z, se in r["windows"]:
            log(f"  {a:4d}  {me:.4f}  {mo:.4f}  {z:+.2f}  {se:.4f}")
        if args.out:
            with open(os.path.join(ROOT, args.out) if not os.path.isabs(args.out) else args.out, "w") as fh:
                fh.write("\n".join(lines) + "\n")
        return
    kw = dict(B=args.batch, steps_per_stage=args.steps_per_stage, n_tasks=args.tasks, n_groups=args.groups, lr=args.lr, n_eval=args.eval)
    log("\n## dropout off: engine and oracle on the same batches from the same weights")
    r = T.run_pair(dev, ocfg, dropout=0.0, log=log, **kw)
    s = T.summarize_pair(r)
    log(f"steps {s['steps']}: max |loss_engine - loss_oracle| {s['max_dloss']:.4f} (first 50 steps {s['max_dloss_first50']:.4f}), mean {s['mean_dloss']:.4f}, "
        f"mean of the last 50 steps {s['dloss_last50_mean']:.4f}")
    log(f"loss first step engine / oracle {s['loss_first'][0]:.4f} / {s['loss_first'][1]:.4f}; mean of the last 10 steps {s['loss_last10'][0]:.4f} / {s['loss_last10'][1]:.4f}")
    log(f"prototype-index agreement per step (each side retrieves with its own weights and prototypes): Q {s['idx_agree_q']:.4f}  V {s['idx_agree_v']:.4f}")
    log(f"held-out greedy answers ({s['heldout']} questions, all tasks and groups): accuracy engine {100 * s['acc_engine']:.2f} %  oracle {100 * s['acc_oracle']:.2f} %  "
        f"(difference {100 * (s['acc_engine'] - s['acc_oracle']):+.2f} points); identical answers engine vs oracle {100 * s['answer_agreement']:.2f} %")
    log(f"wall: engine {s['wall_engine_s']:.1f} s, oracle {s['wall_oracle_s']:.1f} s for the {s['steps']} steps")
    stage = args.steps_per_stage
    log("per stage (task, group): mean |dloss|, mean loss engine / oracle")
    for k in range(0, s["steps"], stage):
        le, lo = r["losses"][0][k:k + stage], r["losses"][1][k:k + stage]
        task, group = r["plan"][k][0], r["plan"][k][1]
        log(f"  task {task} group {group}: {sum(abs(a - b) for a, b in zip(le, lo)) / len(le):.4f}   {sum(le) / len(le):.4f} / {sum(lo) / len(lo):.4f}")
    if args.seeds > 0:
        log(f"\n## dropout 0.1: {args.seeds} seeds per side (the engine's counter-hash masks and torch's RNG differ by construction)")
        ce, co, acc_e, acc_o = [], [], [], []
        for seed in range(args.seeds):
            re_ = T.run_pair(dev, ocfg, dropout=0.1, seed=seed, sides=("engine",), **kw)
            ro_ = T.run_pair(dev, ocfg, dropout=0.1, seed=seed, sides=("oracle",), **kw)
            ce.append(re_["losses"][0])
            co.append(ro_["losses"][0])
            acc_e.append(sum(a == t for a, t in zip(re_["answers"][0], re_["truth"])) / len(re_["truth"]))
            acc_o.append(sum(a == t for a, t in zip(ro_["answers"][0], ro_["truth"])) / len(ro_["truth"]))
            log(f"  seed {seed}: last-10 loss engine {sum(ce[-1][-10:]) / 10:.4f} oracle {sum(co[-1][-10:]) / 10:.4f}; held-out accuracy engine {100 * acc_e[-1]:.2f} % oracle {100 * acc_o[-1]:.2f} %")
        zs, within = T.compare_seeds(ce, co, window=10)
        log(f"mean loss curves in windows of 10 steps: {100 * within:.1f} % of the {len(zs)} windows within 2 sigma; max |z| {max(abs(z[3]) for z in zs):.2f}")
        log("  window start, mean engine, mean oracle, z, sigma")
        for a, me, mo, z, se in zs:
            log(f"  {a:4d}  {me:.4f}  {mo:.4f}  {z:+.2f}  {se:.4f}")

        def ms(v):
            m = sum(v) / len(v)
            return m, (sum((x - m) ** 2 for x in v) / max(1, len(v) - 1)) ** 0.5
        (me, se), (mo, so) = ms(acc_e), ms(acc_o)
        log(f"held-out accuracy over seeds: engine {100 * me:.2f} +- {100 * se:.2f} %, oracle {100 * mo:.2f} +- {100 * so:.2f} %")
        n = len(acc_e)
        sed = (se ** 2 / n + so ** 2 / n) ** 0.5
        log(f"difference of the means engine - oracle: {100 * (me - mo):+.2f} points, standard error {100 * sed:.2f} ({n} seeds per side, {len(re_['truth'])} held-out questions each)")
    if args.out:
        with open(os.path.join(ROOT, args.out) if not os.path.isabs(args.out) else args.out, "w") as fh:
            fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
