"""GPU tests of the "next" rows through the C ABI: the HBM-resident feature store (bit-exact against the
engine's own rounding of the f32 batch), train/test steps fed from the store, and the dual-level loop driver over the engine
with checkpoint / prototype files round-tripped."""
import os
import random

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from vqacl_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _tiny(dev, dropout=0.0, seed=3):
    from oracle import ref_cpu as R
    from test_gpu_model import make_model
    ocfg = R.tiny_cfg()
    return ocfg, make_model(ocfg, R.init_params(ocfg, seed=seed), dev, dropout=dropout)


def test_feature_store_put_gather_bit_exact(dev):
    from vqacl_amd._lib import Vlt5Error, lib, ptr, stream_ptr
    from vqacl_amd.feed import FeatureStore
    g = torch.Generator().manual_seed(5)
    for V, F, n in ((36, 2048, 37), (16, 64, 5), (32, 2048, 9)):
        store = FeatureStore(64, n_boxes=V, feat_dim=F, device=dev)
        feats = torch.randn(n, V, F, generator=g) * 3
        feats[0, 0, :8] = torch.tensor([0.0, -0.0, 1e-40, 65504.0, 3.3895314e38, float("inf"), 1.00390625, 1.01171875])   # ties, denormal, max
        boxes = torch.rand(n, V, 4, generator=g)
        ids = [f"img{i}" for i in range(n)]
        slots = store.put(ids, feats, boxes, chunk=4)
        assert slots == list(range(n)) and len(store) == n and "img3" in store
        want = feats.to(BF)                                      # torch's cast rounds to nearest even, like the engine's
        assert torch.equal(store.feats[:n].cpu().view(torch.int16), want.view(torch.int16))
        assert torch.equal(store.boxes[:n].cpu(), boxes)
        assert not store.feats[n:].any()
        pick = [ids[i] for i in torch.randperm(n, generator=g).tolist()] + [ids[0], ids[0]]
        ref = store.ref(pick)
        gf, gb = store.gather(ref.slots)
        idx = torch.tensor([int(p[3:]) for p in pick])
        assert torch.equal(gf.cpu().view(torch.int16), want[idx].view(torch.int16)) and torch.equal(gb.cpu(), boxes[idx])
        # overwrite keeps the slot
        store.put([ids[2]], feats[2:3] * 2, boxes[2:3])
        assert store.index[ids[2]] == 2 and torch.equal(store.feats[2].cpu(), (feats[2] * 2).to(BF))
        # a slot outside the store yields a zero row (the host API never produces one)
        bad = torch.tensor([1, 64, -1], dtype=torch.long, device=dev)
        zf, zb = store.gather(bad)
        assert zf[0].any() and not zf[1:].any() and not zb[1:].any()
        with pytest.raises(KeyError):
            store.slots(["nope"])
    with pytest.raises(Vlt5Error):
        FeatureStore(4, feat_dim=20, device=dev)
    full = FeatureStore(2, n_boxes=4, feat_dim=8, device=dev)
    with pytest.raises(Vlt5Error):
        full.put(["a", "b", "c"], torch.zeros(3, 4, 8), torch.zeros(3, 4, 4))
    out = torch.empty(1, 4, 8, dtype=BF, device=dev)
    ob = torch.empty(1, 4, 4, device=dev)
    sl = torch.zeros(1, dtype=torch.long, device=dev)
    assert lib().vlt5_feat_gather(ptr(full.feats), ptr(full.boxes), ptr(sl), 2, ptr(out), ptr(ob), 1, 4, 12, stream_ptr()) == 1002
    assert lib().vlt5_feat_gather(None, ptr(full.boxes), ptr(sl), 2, ptr(out), ptr(ob), 1, 4, 8, stream_ptr()) == 1001


class _Dataset:
    """The slice of h5py's dataset interface the reference's item read uses: `read_direct(dest)` and `ds[()]`."""

    def __init__(self, arr):
        self.arr = arr

    def read_direct(self, dest):
        dest[...] = self.arr

    def __getitem__(self, key):
        assert key == ()
        return self.arr.copy() if self.arr.ndim else self.arr[()]


def test_h5_layout_source_fills_the_store(dev):
    """The reference's file layout ({img_id}/features, boxes in pixels, img_w, img_h: vqa_data_memory.py:166-187) read through
    `H5FeatureSource` into the store (h5py itself is not in this image: the file object is a mapping with its dataset interface)."""
    import numpy as np
    from vqacl_amd.feed import FeatureStore, H5FeatureSource, normalize_boxes
    rng = np.random.default_rng(2)
    ids, f, want = [f"COCO_val2014_{k:012d}" for k in range(7)], {}, {}
    for k, i in enumerate(ids):
        w, h = 640 - 17 * k, 480 + 3 * k
        feats = np.maximum(rng.standard_normal((36, 64)).astype(np.float32), 0)
        boxes = np.sort(rng.random((36, 4)).astype(np.float32), axis=1) * np.array([w, h, w, h], dtype=np.float32)
        f[f"{i}/features"], f[f"{i}/boxes"] = _Dataset(feats), _Dataset(boxes)
        f[f"{i}/img_w"], f[f"{i}/img_h"] = _Dataset(np.array(w)), _Dataset(np.array(h))
        want[i] = (torch.from_numpy(feats).to(BF), normalize_boxes(boxes, np.int64(w), np.int64(h)))
    store = FeatureStore(8, n_boxes=36, feat_dim=64, device=dev)
    H5FeatureSource(f, n_boxes=36, feat_dim=64).fill(store, ids, chunk=3)
    assert len(store) == 7
    gf, gb = store.gather(store.slots(ids[::-1]))
    for r, i in enumerate(ids[::-1]):
        assert torch.equal(gf[r].cpu(), want[i][0]) and torch.equal(gb[r].cpu(), want[i][1])
        assert float(gb[r].max()) <= 1.0


def test_real_hdf5_file_fills_the_store(dev):
    """A REAL HDF5 file in the reference's layout (tests/golden/g10_feature_file.h5: written by libhdf5, the library h5py wraps) read by
    path -- h5py when installed, otherwise vqacl_amd.hdf5_io -- into the HBM store at the reference's feature width."""
    import os
    import numpy as np
    from vqacl_amd import hdf5_io
    from vqacl_amd.feed import FeatureStore, H5FeatureSource
    if hdf5_io.find_library() is None:
        pytest.skip("libhdf5 not on this machine")
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ref = np.load(os.path.join(gold, "g10_feature_items.npz"))
    ids = ["9", "COCO_val2014_000000000042", "458752"]
    store = FeatureStore(4, n_boxes=36, feat_dim=2048, device=dev)
    src = H5FeatureSource(os.path.join(gold, "g10_feature_file.h5"))
    src.fill(store, ids, chunk=2)
    gf, gb = store.gather(store.slots(ids))
    for r, i in enumerate(ids):
        feats, _ = src.read(i)
        assert torch.equal(gf[r].cpu(), feats.to(BF))                                   # the store's rounding = the engine's rounding of an f32 batch
        assert torch.equal(gb[r].cpu(), torch.from_numpy(ref[f"{i}/boxes"]))            # = what the reference's item read returned
        assert abs(float(gf[r].double().sum()) - float(ref[f"{i}/vis_feats_sum"])) < 2e-3 * float(ref[f"{i}/vis_feats_sum"])


def test_train_and_test_step_from_store_equal_the_f32_batch(dev):
    """Feeding a step from the store changes nothing downstream: same encoder states bit for bit, same loss, same tokens."""
    from oracle import ref_cpu as R
    from vqacl_amd.feed import FeatureStore
    ocfg, model = _tiny(dev)
    batch = R.synthetic_batch(ocfg, B=6, L=12, V=36, T=4, seed=11)
    store = FeatureStore(16, n_boxes=36, feat_dim=ocfg.feat_dim, device=dev)
    ids = [f"i{k}" for k in range(6)]
    store.put(ids[::-1], batch["vis_feats"].flip(0), batch["boxes"].flip(0))        # slots in another order than the batch
    fed = {k: v for k, v in batch.items() if k not in ("vis_feats", "boxes")}
    fed["feat_ref"] = store.ref(ids)
    _, m2 = _tiny(dev)                                           # same seed: identical weights
    model.train()
    m2.train()
    a = model.train_step(batch, 0, 0.5, 0.3)
    b = m2.train_step(fed, 0, 0.5, 0.3)
    assert torch.equal(a["encoder_hidden_states"], b["encoder_hidden_states"])
    assert float(a["loss"].detach()) == float(b["loss"].detach())
    a["loss"].backward()
    b["loss"].backward()
    for (n1, p1), (n2, p2) in zip(model.named_parameters(), m2.named_parameters()):
        if p1.grad is None:
            assert p2.grad is None
            continue
        # the token-embedding gradient is scattered with f32 atomics (order-dependent in the last bits), the rest reduces in fixed order
        if "shared" in n1 or "embed_tokens" in n1 or "lm_head" in n1:
            assert torch.allclose(p1.grad, p2.grad, rtol=1e-4, atol=1e-6), n1
        else:
            assert torch.equal(p1.grad, p2.grad), n1
    model.eval()
    m2.eval()
    t1 = model.test_step(batch, max_length=6)["token_ids"]
    t2 = m2.test_step(fed, max_length=6)["token_ids"]
    assert torch.equal(t1, t2)
    bad = dict(fed)
    bad["feat_ref"] = store.ref(ids)._replace(slots=store.slots(ids).cpu())
    from vqacl_amd._lib import Vlt5Error
    with pytest.raises(Vlt5Error):
        m2.test_step(bad)


class _Loader(list):
    def __init__(self, batches, n_items):
        super().__init__(batches)
        self.dataset = range(n_items)


def test_continual_trainer_over_engine_and_checkpoint_round_trip(dev, tmp_path):
    """Two tasks x five category groups on the tiny model, fed from the store, rehearsal steps interleaved; the loss goes down on
    the repeated batches, `{task}_LAST.pth` / `Q_prototype.pt` are written in the reference's format and load back bit for bit."""
    from oracle import ref_cpu as R
    from vqacl_amd import checkpoint as CK
    from vqacl_amd.feed import FeatureStore, collate
    from vqacl_amd.loop import CATEGORY_SPLITS, ContinualTrainer
    ocfg, model = _tiny(dev, dropout=0.1)
    tasks = ["q_recognition", "q_location"]
    rng = random.Random(3)
    g = torch.Generator().manual_seed(3)
    n_img = 40
    store = FeatureStore(n_img, n_boxes=36, feat_dim=ocfg.feat_dim, device=dev)
    img_ids = [f"img{k}" for k in range(n_img)]
    store.put(img_ids, torch.relu(torch.randn(n_img, 36, ocfg.feat_dim, generator=g)), torch.rand(n_img, 36, 4, generator=g).sort(-1).values)
    img_cate = {i: CATEGORY_SPLITS[f"G{k % 5 + 1}"][k % 16] for k, i in enumerate(img_ids)}

    def question(task, k):
        n_in, n_tg = rng.randint(4, 12), rng.randint(2, 4)
        return {"img_id": img_ids[k % n_img], "img_cate": img_cate[img_ids[k % n_img]], "question_id": tasks.index(task) * 1000 + k,
                "ques_label": tasks.index(task), "sent": f"{task} {k}", "input_ids": torch.randint(2, ocfg.vocab_size, (n_in,), generator=g),
                "input_length": n_in, "target_ids": torch.cat([torch.randint(2, ocfg.vocab_size, (n_tg - 1,), generator=g), torch.tensor([1])]),
                "target_length": n_tg, "answer": "x", "score": 1.0, "label": {"x": 1.0}}
    pools = {t: [question(t, k) for k in range(40)] for t in tasks}

    def make_loaders(task, kind, exemplars):
        out = {}
        for grp, cats in CATEGORY_SPLITS.items():
            items = exemplars if kind == "memory" else pools[task]
            mine = [q for q in items if q["img_cate"] in cats]
            out[grp] = _Loader([collate(mine[a:a + 4], store=store) for a in range(0, len(mine), 4)], len(mine))
        return out

    events = []
    tr = ContinualTrainer(model, make_loaders, task_items=lambda t: list(pools[t]), img_cate_map=img_cate, task_list=tasks, epochs=3,
                          batch_size=4, lr=2e-3, m_size=10, rng=rng, on_event=lambda kind, **info: events.append((kind, info)),
                          save=lambda name: CK.save_checkpoint(model, str(tmp_path), name))
    tr.train()
    CK.save_prototypes(model, str(tmp_path))
    epochs = [i for k, i in events if k == "epoch"]
    assert len(epochs) == 2 * 5 * 3 and all(e["loss"] == e["loss"] for e in epochs)            # finite everywhere
    first = [e for e in epochs if e["task"] == tasks[0] and e["epoch"] == 0]
    last = [e for e in epochs if e["task"] == tasks[0] and e["epoch"] == 2]
    assert sum(e["loss"] for e in last) < sum(e["loss"] for e in first)
    steps = [i["source"] for k, i in events if k == "step"]
    assert "memory" in steps and all(i["source"] == "new" for k, i in events if k == "step" and i["task_idx"] == 0)
    assert [i["each_memory"] for k, i in events if k == "memory"] == [10]
    assert float(model.Q_prototype.abs().sum()) > 0
    sd = torch.load(tmp_path / "q_location_LAST.pth")
    assert all(k.startswith("module.") for k in sd) and "module.shared.weight" in sd and "module.lm_head.weight" in sd
    _, fresh = _tiny(dev, seed=99)
    res = CK.load_checkpoint(fresh, str(tmp_path / "q_location_LAST"), map_location="cpu")
    assert not res.unexpected_keys
    CK.load_prototypes(fresh, str(tmp_path))
    a, b = model.state_dict(), fresh.state_dict()
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    assert torch.equal(fresh.Q_prototype, model.Q_prototype) and torch.equal(fresh.V_prototype, model.V_prototype)
    # the reloaded model continues identically: one eval-mode forward gives the same logits
    batch = make_loaders(tasks[0], "train", [])["G1"][0]
    model.eval()
    fresh.eval()
    assert torch.equal(model.test_step(batch, max_length=5)["token_ids"], fresh.test_step(batch, max_length=5)["token_ids"])


def test_predict_and_score_with_the_real_model(dev):
    """`loop.predict` -> `VLT5VQA.test_step` (greedy decoding with the key/value cache, fed from the store) -> answers through a
    tokenizer -> `VQAEvaluator`: the evaluation path of `Trainer.test` end to end on the tiny model."""
    from types import SimpleNamespace
    from vqacl_amd.evaluate import VQAEvaluator
    from vqacl_amd.feed import FeatureStore, collate
    from vqacl_amd.loop import evaluate, predict
    ocfg, model = _tiny(dev)
    g = torch.Generator().manual_seed(8)

    class Tok:                                               # token id -> word; specials (pad 0, eos 1) dropped like skip_special_tokens
        def batch_decode(self, ids, skip_special_tokens=True):
            return [" ".join(f"w{int(t)}" for t in row if int(t) > 1) for row in ids.tolist()]
    model.tokenizer = Tok()
    store = FeatureStore(8, n_boxes=36, feat_dim=ocfg.feat_dim, device=dev)
    imgs = [f"im{k}" for k in range(8)]
    store.put(imgs, torch.relu(torch.randn(8, 36, ocfg.feat_dim, generator=g)), torch.rand(8, 36, 4, generator=g).sort(-1).values)
    items = [{"img_id": imgs[k % 8], "img_cate": k % 80, "question_id": 500 + k, "ques_label": 0, "sent": f"q{k}",
              "input_ids": torch.randint(2, ocfg.vocab_size, (5 + k % 4,), generator=g), "input_length": 5 + k % 4} for k in range(10)]
    loader = [collate(items[a:a + 4], store=store) for a in range(0, 10, 4)]
    ans = predict(model, loader)
    assert sorted(ans) == [500 + k for k in range(10)] and all(isinstance(a, str) for a in ans.values())
    assert ans == predict(model, loader)                     # greedy decoding is deterministic
    # score against "annotations" built from the model's own answers for half of the questions
    gt = {q: {"answers": [{"answer": (a if q % 2 else "zzz"), "answer_id": i} for i in range(10)], "question_type": "t", "answer_type": "other"}
          for q, a in ans.items()}
    ds = SimpleNamespace(id2datum={q: {"label": {a: 1.0}} for q, a in ans.items()}, id2datum_gt=gt)
    acc = evaluate(model, loader, VQAEvaluator(ds))
    assert acc["overall"] == 50.0 and acc["topk_score"] == 1.0
