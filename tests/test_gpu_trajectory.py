"""Accuracy proxy, short form (the full figures: tools/trajectory.py -> profiles/r05_trajectory.txt): VL-T5-base at the benched batch
size trained by the engine and by the fp32 oracle (torch eager on the same GPU) side by side through the dual-level continual schedule
of Trainer.train (vqacl.py:314-373) -- three question-type tasks x two category groups, a new AdamW + warm-up per (task, group),
rehearsal batches from the second task on, clip 5 -- on a synthetic VQA problem with a learnable rule.  The two trajectories are chaotic
copies of each other (bf16 operands against fp32), so the thresholds are statistical: stated below, measured values in the parity log."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_engine_and_oracle_trained_side_by_side_agree():
    from oracle import ref_cpu as R
    from test_gpu_model import parity_log
    import trajectory_lib as T
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    ocfg = R.Cfg(dropout=0.0)
    r = T.run_pair(dev, ocfg, dropout=0.0, B=80, steps_per_stage=40, n_tasks=3, n_groups=2, n_eval=240)
    s = T.summarize_pair(r)
    parity_log(f"trajectory (base, B=80, {s['steps']} optimizer steps, 3 tasks x 2 groups, rehearsal): max |dloss| {s['max_dloss']:.4f} "
               f"(first 50: {s['max_dloss_first50']:.4f}), mean {s['mean_dloss']:.4f}; prototype-index agreement Q {s['idx_agree_q']:.3f} V {s['idx_agree_v']:.3f}; "
               f"held-out accuracy engine {100 * s['acc_engine']:.1f} % oracle {100 * s['acc_oracle']:.1f} %, identical answers {100 * s['answer_agreement']:.1f} %")
    # the same weights and batches: the first stage tracks the oracle closely (measured 0.009 over 50 steps at 50 steps per stage) ...
    assert s["max_dloss_first50"] < 0.05, s
    # ... later the two runs are different samples of the same training process: the mean gap stays small against losses of O(1)
    assert s["mean_dloss"] < 0.1, s
    assert s["idx_agree_q"] > 0.85 and s["idx_agree_v"] > 0.9, s
    # both learn the rule to the same degree: held-out accuracy within 12 points of each other (240 questions: sigma ~ 3 points per side) and
    # both clearly above chance (a random answer sequence is right with probability < 1 / 48)
    assert abs(s["acc_engine"] - s["acc_oracle"]) < 0.12 and min(s["acc_engine"], s["acc_oracle"]) > 0.05, s
    # the training loss went down on both sides (the last ten steps of the last stage against the first step), to the same level
    assert s["loss_last10"][0] < 0.25 * s["loss_first"][0] and s["loss_last10"][1] < 0.25 * s["loss_first"][1], s
    assert abs(s["loss_last10"][0] - s["loss_last10"][1]) < 0.25 * max(s["loss_last10"]) + 0.05, s
