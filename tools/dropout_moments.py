#!/usr/bin/env python3
"""Dropout ON against the oracle without a shared random stream: moments over dropout seeds, both sides from the same weights on the
same batch (tests/trajectory_lib.py: dropout_moments).  The oracle runs as torch eager fp32 on the same GPU.

    python tools/dropout_moments.py --out profiles/rNN_dropout_moments.txt       # VL-T5-base, B = 80, 48 seeds per side, at the initialisation and after 60 steps
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
    ap.add_argument("--seeds", type=int, default=48)
    ap.add_argument("--warm", type=int, nargs="*", default=[0, 60])
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    from oracle import ref_cpu as R
    from tests import trajectory_lib as T
    from vqacl_amd.build import source_hash
    dev = torch.device("cuda", 0)
    ocfg = R.tiny_cfg(vocab_size=3200, feat_dim=256) if args.tiny else R.Cfg(dropout=0.0)
    lines = []

    def log(x=""):
        print(x, flush=True)
        lines.append(x)
    log(f"# dropout moments: {'tiny' if args.tiny else 'VL-T5-base'}; the engine's counter-hash masks against torch's generator, {args.seeds} seeds per side; "
        "what has to agree is the distribution over the seeds")
    log(f"# source_sha16 {source_hash()}   {time.strftime('%Y-%m-%d %H:%M:%S')}   checker: oracle/ref_cpu.py as torch eager fp32 on the same GPU")
    for w in args.warm:
        t0 = time.time()
        r = T.dropout_moments(dev, ocfg, B=args.batch, K=args.seeds, warm_steps=w)
        log("")
        for ln in T.format_moments(r):
            log(ln)
        log(f"  ({time.time() - t0:.0f} s)")
    if args.out:
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
