#!/usr/bin/env python3
"""Soak run of the train step: N optimizer steps at the benched shape (B = 80, store-fed, dropout on) twice from the same seed -- the loss
curve must be finite, fall, and be bit-identical between the two runs; device memory must not grow; a greedy decode at the end.
usage: python tools/soak.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch  # noqa: E402
from vqacl_amd import FusedAdamW, VLT5Config, VLT5VQA, reference_param_groups  # noqa: E402
from vqacl_amd.feed import FeatureStore  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda")
B, L, V, T = 80, 20, 36, 5


def run():
    cfg = VLT5Config(dropout_rate=0.1)
    torch.manual_seed(66666)
    model = VLT5VQA(cfg, device=dev)
    model.train()
    opt = FusedAdamW(reference_param_groups(model, 0.01), model, lr=1e-4, eps=1e-6, max_grad_norm=5.0)
    store = FeatureStore(1024, n_boxes=V, feat_dim=cfg.feat_dim, device=dev)
    gen = torch.Generator(device=dev).manual_seed(66666)
    for a in range(0, 1024, 256):
        store.put(list(range(a, a + 256)), torch.relu(torch.randn(256, V, cfg.feat_dim, device=dev, generator=gen)) * 1.5,
                  torch.rand(256, V, 4, device=dev, generator=gen).sort(-1).values)
    hg = torch.Generator().manual_seed(1234)
    batches = []
    for i in range(16):                                      # 16 fixed batches, cycled: the loss must fall on them
        small = synthetic_batch(B, L, V, T, seed=66666 + i, with_feats=False)
        small["img_ids"] = torch.randint(0, 1024, (B,), generator=hg).tolist()
        batches.append(small)
    losses, mem = [], []
    t0 = time.perf_counter()
    for i in range(N):
        b = batches[i % 16]
        fed = {k: v for k, v in b.items() if k != "img_ids"}
        fed["feat_ref"] = store.ref(b["img_ids"])
        res = model.train_step(fed, i % 3, 0.5, 0.3)
        res["loss"].backward()
        opt.step()
        for p in model.parameters():
            p.grad = None
        if i % 10 == 9 or i == 0:
            losses.append(float(res["loss"].detach()))
            mem.append(torch.cuda.memory_allocated())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    model.eval()
    fb = store.ref(batches[0]["img_ids"])
    toks = model.greedy_generate(batches[0]["input_ids"].to(dev), fb, max_length=8)
    return losses, mem, dt, toks.cpu(), model.flat_params().float().cpu().clone()


a = run()
b = run()
print(f"{N} steps in {a[2]:.2f} s ({a[2] / N * 1e3:.2f} ms per step incl. the loss read every 10 steps); loss {a[0][0]:.4f} -> {a[0][-1]:.4f}")
print("loss curve:", " ".join(f"{x:.3f}" for x in a[0][::3]))
assert all(x == x and abs(x) < 1e4 for x in a[0]), "non-finite loss"
assert a[0][-1] < 0.7 * a[0][0], "the loss did not fall"
assert a[0] == b[0], "two runs from the same seed differ"
assert torch.equal(a[4], b[4]), "parameters differ between the two runs"
assert torch.equal(a[3], b[3]), "greedy tokens differ between the two runs"
print(f"two runs bit-identical (loss curve, {a[4].numel()} parameters, greedy tokens); device memory {a[1][1] / 2**20:.0f} MiB after step 10, "
      f"{a[1][-1] / 2**20:.0f} MiB after step {N}")
assert a[1][-1] <= a[1][1] * 1.01 + (1 << 20), "device memory grew"
print("soak ok")
