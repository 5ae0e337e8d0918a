"""CPU tests of the host-side "next" rows (SURVEY 8 f-2 feed, f-3 loop driver, f-4 evaluator/checkpoint) against golden vectors
produced by the reference's own code (oracle/make_golden_host.py): collate and box normalisation bit-exact, answer
normalisation string-exact, accuracies and continual-learning metrics equal, rehearsal-memory indices bit-exact."""
import copy
import json
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN


# ------------------------------------------------------------------------------------------------ f-2
def test_collate_matches_reference_collate_fn():
    from vqacl_amd.feed import collate
    G = torch.load(os.path.join(GOLDEN, "g7_feed.pt"), weights_only=False)
    for case in G["collate"]:
        got = collate(copy.deepcopy(case["entries"]), pad_token_id=0)
        want = case["batch"]
        got.pop("args")
        assert set(got) == set(want), set(got) ^ set(want)
        for k, v in want.items():
            if torch.is_tensor(v):
                assert got[k].dtype == v.dtype and got[k].shape == v.shape, k
                assert torch.equal(got[k], v), k
            else:
                assert got[k] == v, k
    one = collate(copy.deepcopy(G["collate"][0]["entries"]))
    assert (one["target_ids"] == 0).sum() == 0 and (one["target_ids"] == -100).any()       # every pad id became -100
    assert one["cate_labels"].sum(1).eq(1).all() and one["ques_labels"].shape[1] == 10


def test_box_normalisation_bit_exact_and_asserts():
    from vqacl_amd.feed import normalize_boxes
    G = torch.load(os.path.join(GOLDEN, "g7_feed.pt"), weights_only=False)
    for case in G["boxes"]:
        got = normalize_boxes(case["raw"].numpy(), np.array(case["img_w"])[()], np.array(case["img_h"])[()])
        assert got.dtype == torch.float32 and torch.equal(got, case["boxes"])
        assert float(got.max()) <= 1.0 and float(got.min()) >= 0.0
    bad = G["boxes"][0]["raw"].numpy().copy()
    bad[3, 2] = G["boxes"][0]["img_w"] * 1.01
    with pytest.raises(AssertionError):
        normalize_boxes(bad, G["boxes"][0]["img_w"], G["boxes"][0]["img_h"])
    bad[3, 2] = -5.0
    with pytest.raises(AssertionError):
        normalize_boxes(bad, G["boxes"][0]["img_w"], G["boxes"][0]["img_h"])


def test_feature_store_needs_gpu_and_library():
    from vqacl_amd import _lib
    from vqacl_amd.feed import FeatureStore
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with pytest.raises(_lib.Vlt5Error):
        FeatureStore(8)


# ------------------------------------------------------------------------------------------------ f-4
def _g8():
    return json.load(open(os.path.join(GOLDEN, "g8_evaluator.json")))


def test_answer_normalisation_string_exact():
    from vqacl_amd.evaluate import normalize_answer, VQAEvaluator
    G = _g8()
    assert len(G["answers"]) > 300
    for a, want in zip(G["answers"], G["normalized"]):
        assert normalize_answer(a) == want, (a, normalize_answer(a), want)
    assert VQAEvaluator().normalize_answer("The two   Frisbees!") == normalize_answer("The two   Frisbees!") == "2 frisbees"


def test_vqa_accuracy_matches_reference_evaluator():
    from types import SimpleNamespace
    from vqacl_amd.evaluate import VQAEvaluator
    G = _g8()
    id2datum = {int(k): v for k, v in G["id2datum"].items()}
    gt = {int(k): v for k, v in G["gt"].items()}
    pred = {int(k): v for k, v in G["pred"]}                       # ordered pairs: the evaluator sums in iteration order
    ev = VQAEvaluator(SimpleNamespace(id2datum=id2datum, id2datum_gt=gt))
    assert ev.evaluate(pred) == pytest.approx(G["expected"]["topk"], abs=1e-12)
    # same call order as the generating script: the in-place normalisation of the human answers carries over between calls
    for tag, flag in (("all", None), ("optimal", True), ("not_optimal", False)):
        got = ev.evaluate_raw(pred, is_topk_optimal=flag)
        assert got == G["expected"]["raw_" + tag], tag
    assert {str(k): v for k, v in ev.evalQA.items()} == G["expected"]["evalQA_last"]
    assert ev.evaluate_raw({}) == {"overall": 0, "perQuestionType": {}, "perAnswerType": {}}


def test_continual_metrics_match_reference():
    from vqacl_amd.evaluate import evaluate_metric, result_matrix
    for case in _g8()["metrics"]:
        results = {a: dict(row) for a, row in case["results"]}          # ordered pairs: the task order is the dict order
        got = evaluate_metric(results, case["start"])
        for k, want in case["metric"].items():
            assert got[k] == pytest.approx(want, abs=1e-9), (k, got[k], want)
    r = {"a": {"a": 50.0, "b": 0.0}, "b": {"a": 40.0, "b": 60.0}}
    assert result_matrix(r) == [[50.0, -1.0], [40.0, 60.0]]
    m = evaluate_metric(r, all_tasks=["a", "b"], comp_tasks=["b"])
    assert m["Avg_acc"] == 50.0 and m["Avg_forget"] == 10.0 and m["Incre_avg_acc_6Q"] == [-1, 60.0]


def test_checkpoint_key_mapping_and_files(tmp_path):
    from vqacl_amd import checkpoint as CK

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = torch.nn.Linear(3, 2)
            self.Q_prototype = torch.zeros(10, 4)
            self.V_prototype = torch.zeros(80, 4)

    m = Tiny()
    path = CK.save_checkpoint(m, str(tmp_path), "q_count_LAST")
    sd = torch.load(path)
    assert path.endswith("q_count_LAST.pth") and set(sd) == {"module.encoder.weight", "module.encoder.bias"}
    m2 = Tiny()
    res = CK.load_checkpoint(m2, os.path.join(str(tmp_path), "q_count_LAST"))          # the reference passes no extension
    assert not res.missing_keys and not res.unexpected_keys and torch.equal(m2.encoder.weight, m.encoder.weight)
    legacy = {"module.vis_encoder.weight": torch.ones(2, 3), "module.model.vis_encoder.bias": torch.ones(2), "plain": torch.ones(1)}
    assert set(CK.from_reference_keys(legacy)) == {"encoder.weight", "encoder.bias", "plain"}
    m.Q_prototype += 1.5
    CK.save_prototypes(m, str(tmp_path))
    CK.load_prototypes(m2, str(tmp_path))
    assert torch.equal(m2.Q_prototype, m.Q_prototype) and tuple(torch.load(tmp_path / "V_prototype.pt").shape) == (80, 4)


# ------------------------------------------------------------------------------------------------ f-3
def _g9():
    return json.load(open(os.path.join(GOLDEN, "g9_loop.json")))


def test_rehearsal_memory_indices_bit_exact():
    from vqacl_amd.loop import ALL_TASKS, ExemplarMemory
    G = _g9()
    for case in G["cases"]:
        mem = ExemplarMemory(case["M"])
        rng = random.Random()
        rng.seed(case["seed"])
        for step in case["steps"]:
            t = step["task_idx"]
            items = copy.deepcopy(G["pools"][ALL_TASKS[t - 1]])            # the reference re-reads the JSON, then shuffles it
            allx, each = mem.update(t, items, G["img_cate"], rng)
            assert each == step["each_memory"]
            assert [d["question_id"] for d in allx] == step["all"]
            assert {g: [[d["question_id"] for d in ts] for ts in mem.sets[g]] for g in mem.sets} == step["sets"]
    with pytest.raises(ValueError):
        ExemplarMemory(10).update(0, [], {})


def test_group_order_and_warmup_arithmetic():
    from vqacl_amd.loop import constant_schedule_with_warmup, shuffled_groups, warmup_iters
    G = _g9()
    rng = random.Random()
    rng.seed(77)
    assert [shuffled_groups(rng=rng) for _ in range(6)] == G["group_orders_seed77"]
    for w in G["warmup"]:
        assert warmup_iters(w["total"], w["batch_size"], w["epochs"], w["ratio"]) == w["warmup_iters"]
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1e-4)
    sch = constant_schedule_with_warmup(opt, 4)
    lrs = []
    for _ in range(6):
        lrs.append(sch.get_last_lr()[0])
        opt.step()
        sch.step()
    assert lrs == pytest.approx([0.0, 0.25e-4, 0.5e-4, 0.75e-4, 1e-4, 1e-4])


class _Loader(list):
    def __init__(self, batches, n_items):
        super().__init__(batches)
        self.dataset = range(n_items)


class _FakeModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(1))
        self.calls = []

    def train_step(self, batch, task_idx, alpha, beta, each_memory, total):
        self.calls.append((batch["tag"], task_idx, each_memory))
        return {"loss": (self.w * batch["x"]).sum()}


def test_dual_level_schedule_order():
    """Order of operations of Trainer.train (vqacl.py:147-424): tasks in order; per task the memory is rebuilt from the previous
    task, the 5 groups run in a fresh shuffled order with a fresh optimizer + warm-up each, the composition group is skipped for
    every task but the first, a new-data step is followed by a rehearsal step once the memory is non-empty, checkpoint + test
    after each task."""
    from vqacl_amd.loop import CATEGORY_SPLITS, ContinualTrainer
    tasks = ["q_recognition", "q_location", "q_judge"]
    img_cate = {f"i{k}": CATEGORY_SPLITS[f"G{k % 5 + 1}"][0] for k in range(50)}
    pools = {t: [{"img_id": f"i{k}", "question_id": k} for k in range(50)] for t in tasks}
    events, model = [], _FakeModel()

    def make_loaders(task, kind, exemplars):
        out = {}
        for g in CATEGORY_SPLITS:
            if kind == "memory":
                mine = [e for e in exemplars if img_cate[e["img_id"]] in CATEGORY_SPLITS[g]]
                out[g] = _Loader([{"tag": f"mem:{g}", "x": torch.ones(1)}] if mine else [], len(mine))
            else:
                out[g] = _Loader([{"tag": f"{kind}:{task}:{g}:{b}", "x": torch.ones(1)} for b in range(3)], 3 * 4)
        return out

    rng = random.Random(5)
    tr = ContinualTrainer(model, make_loaders, task_items=lambda t: copy.deepcopy(pools[t]), img_cate_map=img_cate, task_list=tasks,
                          epochs=2, batch_size=4, m_size=20, comp_cate="G3", rng=rng,
                          make_optimizer=lambda m, lr: torch.optim.SGD(m.parameters(), lr=lr),
                          on_event=lambda kind, **info: events.append((kind, info)), save=lambda name: events.append(("save", name)),
                          test=lambda task: events.append(("test", task)), evaluate=lambda loader: events.append(("eval", len(loader))))
    tr.train()
    kinds = [k for k, _ in events]
    assert [i["task"] for k, i in events if k == "task"] == tasks
    assert [i for k, i in events if k == "save"] == [t + "_LAST" for t in tasks]
    assert [i["each_memory"] for k, i in events if k == "memory"] == [20, 10]            # int(M / task_idx)
    assert [i["size"] for k, i in events if k == "memory"] == [20, 20]                    # 5 groups x share 4, then 2 tasks x 5 x 2
    # group order: a fresh shuffle per task from the same rng stream the memory update draws from
    check = random.Random(5)
    for ti, t in enumerate(tasks):
        if ti > 0:
            check.shuffle(copy.deepcopy(pools[tasks[ti - 1]]))
        keys = list(CATEGORY_SPLITS)
        check.shuffle(keys)
        assert [i["group"] for k, i in events if k == "group" and i["task"] == t] == keys
    assert [(i["task"], i["group"]) for k, i in events if k == "skip"] == [("q_location", "G3"), ("q_judge", "G3")]
    # first task: no memory -> 5 groups x 2 epochs x 3 batches of new data only; total_train_num = dataset size
    first = model.calls[:30]
    assert all(tag.startswith("train:q_recognition") and ti == 0 and em == 0 for tag, ti, em in first)
    g0 = [i for k, i in events if k == "group"][0]
    assert g0["total_train_num"] == 12 and g0["warmup_iters"] == int(int(12 / 4) * 2 * 0.05)
    # later tasks: new/memory alternate, 4 trained groups x 2 epochs x 3 pairs, memory batch cycled
    second = model.calls[30:30 + 48]
    assert [c[0].split(":")[0] for c in second] == ["train", "mem"] * 24
    assert all(ti == 1 and em == 20 for _, ti, em in second)
    g5 = [i for k, i in events if k == "group"][5]
    assert g5["total_train_num"] == 24                                                    # doubled when the group has rehearsal data
    assert kinds.count("eval") == (5 + 4 + 4) * 2 and kinds.count("test") == 3
    assert float(tr.task_total_num[1]) == 60.0 and all(p.grad is None for p in model.parameters())


def test_predict_evaluate_and_result_matrix():
    """`Trainer.predict` / `evaluate` / the test pass (vqacl.py:527-631) over a stub model: answers keyed by question id, raw
    accuracy + topk score, result matrix filled for the tasks trained so far only, then the continual metrics on it."""
    from types import SimpleNamespace
    from vqacl_amd.evaluate import VQAEvaluator, evaluate_metric
    from vqacl_amd.loop import evaluate, predict, test_seen_tasks

    class Stub:
        def __init__(self, answers):
            self.answers, self.mode = answers, None

        def eval(self):
            self.mode = "eval"

        def test_step(self, batch):
            return {"pred_ans": [self.answers[q] for q in batch["question_ids"]]}

    def dataset(qids, truth):
        id2datum = {q: {"label": {truth[q]: 1.0}} for q in qids}
        gt = {q: {"answers": [{"answer": truth[q], "answer_id": k} for k in range(10)], "question_type": "what", "answer_type": "other"}
              for q in qids}
        return SimpleNamespace(id2datum=id2datum, id2datum_gt=gt)

    truth = {q: ("yes" if q % 2 else "2 dogs") for q in range(12)}
    model = Stub({q: ("yes" if q % 2 else ("Two dogs!" if q < 8 else "cat")) for q in range(12)})
    loaders = {"q_recognition": [{"question_ids": [0, 1, 2]}, {"question_ids": [3, 4, 5]}], "q_location": [{"question_ids": [6, 7, 8, 9]}],
               "q_judge": [{"question_ids": [10, 11]}]}
    evals = {t: VQAEvaluator(dataset([q for b in l for q in b["question_ids"]], truth)) for t, l in loaders.items()}
    got = predict(model, loaders["q_recognition"])
    assert model.mode == "eval" and got == {q: model.answers[q] for q in range(6)}
    acc = evaluate(model, loaders["q_location"], evals["q_location"])
    assert acc["overall"] == 75.0 and acc["topk_score"] == 0.5          # "Two dogs!" normalises to "2 dogs": counted by the raw score, not by the exact-match topk score
    tasks = ["q_recognition", "q_location", "q_judge"]
    matrix = {}
    test_seen_tasks(model, "q_recognition", tasks, {"q_recognition": 1, "q_location": 0, "q_judge": 0}, loaders, evals, matrix)
    test_seen_tasks(model, "q_location", tasks, {"q_recognition": 1, "q_location": 1, "q_judge": 0}, loaders, evals, matrix)
    assert matrix == {"q_recognition": {"q_recognition": 100.0}, "q_location": {"q_recognition": 100.0, "q_location": 75.0}}
    m = evaluate_metric(matrix)
    assert m["Avg_acc"] == 87.5 and m["Avg_forget"] == 0.0
