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


@pytest.mark.timeout(600)
def test_dropout_on_matches_the_oracle_in_distribution():
    """The benched configuration has dropout ON, and the engine's counter-hash masks share no random stream with torch's generator:
    element-wise parity is undefined there.  What is defined is the distribution over the seeds.  Same weights (30 dropout-off steps
    away from the initialisation), same batch, 16 seeds per side (VL-T5-base, B = 32): the per-element variance of the encoder output,
    the decoder output and the logits summed over the elements must agree to 3 % (measured 0.999 / 1.001 / 1.001 -- the oracle with ONE
    of its dropout sites removed: 0.57-0.99, most sites 0.93-0.97), the mean loss to 4 standard errors, and the mean GRADIENT of the
    two sides must be as close to each other as the seed noise in it allows -- closer than the two halves of either side's seeds are."""
    from oracle import ref_cpu as R
    from test_gpu_model import parity_log
    import trajectory_lib as T
    dev = torch.device("cuda", 0)
    r = T.dropout_moments(dev, R.Cfg(dropout=0.0), B=32, K=16, warm_steps=30)
    k = r["keys"]
    parity_log("dropout 0.1 in distribution (base, B=32, 16 seeds per side): variance engine/oracle enc {:.4f} dec {:.4f} logits {:.4f} grad {:.4f}; "
               "loss {:.5f} / {:.5f} ({:+.2f} se); cos(mean grad) {:.5f} (halves {:.5f} / {:.5f})".format(
                   k["enc"]["var_ratio"], k["dec"]["var_ratio"], k["logits"]["var_ratio"], k["grad"]["var_ratio"], r["loss"]["engine"], r["loss"]["oracle"],
                   r["loss"]["z"], k["grad"]["cos_means"], k["grad"]["cos_halves_engine"], k["grad"]["cos_halves_oracle"]))
    for name in ("enc", "dec", "logits"):
        assert 0.97 < k[name]["var_ratio"] < 1.03, (name, k[name])
        # the means differ by what 16 seeds leave of the seed noise (+ the bf16 difference), not by more
        assert k[name]["mean_rel_diff"] < 1.15 * k[name]["noise_floor"] + 2 * r["masks_off"][name], (name, k[name], r["masks_off"])
        assert abs(k[name]["norm_ratio"] - 1) < 0.01, (name, k[name])
    assert abs(r["loss"]["z"]) < 4, r["loss"]
    # the masks being there at all: the loss under dropout differs from the masks-off loss by much more than the two sides differ
    assert abs(r["loss"]["engine"] - r["loss"]["masks_off"][0]) > 10 * abs(r["loss"]["engine"] - r["loss"]["oracle"]), r["loss"]
    g = k["grad"]
    assert 0.8 < g["var_ratio"] < 1.25, g              # gradients are heavier-tailed than activations: a wider band
    assert g["cos_means"] > min(g["cos_halves_engine"], g["cos_halves_oracle"]), g
    assert abs(g["norm_ratio"] - 1) < 0.05, g
    assert r["masks_off"]["grad_cos"] > 0.999, r["masks_off"]
