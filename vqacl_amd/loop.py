"""Dual-level continual-learning loop driver and rehearsal-memory builder.

Reference: VL-T5/src/vqacl.py:147-424 (`Trainer.train`), :429-509 (`Trainer.train_step`), Question_type.py:7-24
(task / category-group constants, `random_dic`), src/trainer_base.py:130-198 (optimizer + schedule factory).

Outer level: the 10 question-type tasks in order.  Inner level: the 5 object-category groups in a freshly shuffled order per
task; the optimizer and the warm-up schedule are rebuilt per group; a group equal to `comp_cate` is skipped for every task but
the first (novel-composition protocol); within a group every batch of new data is followed by one batch of rehearsal data
(`zip(train, cycle(memory))`), each an optimizer step of its own.  After a task: checkpoint `{task}_LAST`, then the test pass.

The driver only sequences: the step itself is `VLT5VQA.train_step` + `FusedAdamW` (the HIP engine).  What the reference does
with CWD-relative JSON paths and DataLoaders is behind two callables (`task_items`, `make_loaders`), so the schedule runs
unchanged over the resident feature store (vqacl_amd/feed.py) or any other feed.
"""
import random as _random
from itertools import cycle

import torch

# Question_type.py:16-17 -- the 10 linguistic-driven tasks, and the 6 used by the novel-composition metrics
ALL_TASKS = ["q_recognition", "q_location", "q_judge", "q_commonsense", "q_count", "q_action", "q_color", "q_type",
             "q_subcategory", "q_causal"]
COMP_TASKS = ["q_location", "q_count", "q_action", "q_color", "q_type", "q_subcategory"]
# Question_type.py:20-24 -- the 5 visual-driven groups of 16 COCO object categories (indices into the 80 classes)
CATEGORY_SPLITS = {
    "G1": [58, 48, 55, 36, 64, 1, 70, 73, 42, 15, 6, 18, 49, 59, 31, 2],
    "G2": [19, 77, 22, 9, 24, 53, 12, 13, 78, 50, 47, 41, 32, 28, 54, 23],
    "G3": [60, 8, 34, 25, 67, 4, 14, 68, 3, 79, 0, 5, 65, 20, 71, 39],
    "G4": [35, 29, 66, 40, 43, 26, 72, 10, 38, 61, 76, 44, 75, 69, 16, 57],
    "G5": [45, 33, 63, 56, 21, 11, 62, 74, 17, 52, 46, 30, 27, 51, 37, 7],
}


def shuffled_groups(groups=CATEGORY_SPLITS, rng=_random):
    """`random_dic` (Question_type.py:7-13): the group names in a shuffled order (one `rng.shuffle` of the key list)."""
    keys = list(groups.keys())
    rng.shuffle(keys)
    return keys


class ExemplarMemory:
    """Rehearsal memory of `m_size` questions, split evenly over the tasks seen so far and, within a task, over the category
    groups (vqacl.py:143-145, 165-209).  `sets[group][task]` is a list of the dataset's question dicts."""

    def __init__(self, m_size, groups=CATEGORY_SPLITS):
        self.M = int(m_size)
        self.groups = groups
        self.sets = {g: [] for g in groups}

    def update(self, task_idx, prev_task_items, img_cate_map, rng=_random):
        """Called before training task `task_idx` >= 1 with the training questions of task `task_idx - 1` (a list the call
        shuffles IN PLACE with `rng`, as the reference shuffles what it has just loaded).  Per group: take questions in shuffled
        order whose image category lies in the group until the per-group share is reached, then cut every older task's list to
        the new share.  Returns (all exemplars: groups in dict order, tasks in order; each_memory = int(M / task_idx))."""
        if task_idx < 1:
            raise ValueError("the memory is first built before the second task")
        each_memory = int(self.M / task_idx)
        rng.shuffle(prev_task_items)
        share = int(each_memory / len(self.groups))
        for g, cats in self.groups.items():
            chosen = []
            self.sets[g].append(chosen)
            for item in prev_task_items:
                img = item["img_id"]
                if img in img_cate_map and img_cate_map[img] in cats:
                    chosen.append(item)
                    if len(chosen) >= share:          # checked after the append: a share of 0 still draws one, cut below
                        break
        for g in self.groups:
            for i in range(len(self.sets[g])):
                self.sets[g][i] = self.sets[g][i][:share]
        return self.all(), each_memory

    def all(self):
        out = []
        for g in self.sets:
            for task_set in self.sets[g]:
                out += task_set
        return out


def warmup_iters(total_train_num, batch_size, epochs, warmup_ratio, gradient_accumulation_steps=1):
    """trainer_base.py:138-142."""
    batch_per_epoch = int(total_train_num / batch_size)
    t_total = batch_per_epoch // gradient_accumulation_steps * epochs
    return int(t_total * warmup_ratio)


def constant_schedule_with_warmup(optimizer, n_warmup):
    """transformers `get_constant_schedule_with_warmup` (trainer_base.py:184): lr * min(1, step / max(1, n_warmup))."""
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda step: float(step) / float(max(1.0, n_warmup)) if step < n_warmup else 1.0)


def predict(model, loader, group=None):
    """`Trainer.predict` (vqacl.py:585-620): greedy answers of every batch of `loader` -> {question_id: answer}; under
    torch.distributed the per-rank dicts are gathered and merged (the reference's `dist_utils.all_gather`), every rank returns the
    union."""
    import torch.distributed as dist
    model.eval()
    quesid2ans = {}
    with torch.no_grad():
        for batch in loader:
            results = model.test_step(batch)
            for qid, ans in zip(batch["question_ids"], results["pred_ans"]):
                quesid2ans[qid] = ans
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        parts = [None] * dist.get_world_size(group)
        dist.all_gather_object(parts, quesid2ans, group=group)
        quesid2ans = {}
        for part in parts:
            quesid2ans.update(part)
    return quesid2ans


def evaluate(model, loader, evaluator, group=None):
    """`Trainer.evaluate` (vqacl.py:622-631): the evaluator's raw accuracies plus `topk_score`."""
    quesid2ans = predict(model, loader, group)
    acc = evaluator.evaluate_raw(quesid2ans)
    acc["topk_score"] = evaluator.evaluate(quesid2ans) if quesid2ans else 0.0
    return acc


def test_seen_tasks(model, task, task_list, task_iftrain, test_loaders, evaluators, result_matrix, group=None):
    """The test pass after task `task` (vqacl.py:527-582): every task trained so far is predicted and scored; the overall raw
    accuracy goes into result_matrix[task][test_task] (the matrix `evaluate_metric` reads).  Stops at the first untrained task."""
    result_matrix.setdefault(task, {})
    for test_task in task_list:
        if task_iftrain.get(test_task, 0) == 0:
            break
        acc = evaluators[test_task].evaluate_raw(predict(model, test_loaders[test_task], group))
        result_matrix[task][test_task] = acc["overall"]
    return result_matrix


class ContinualTrainer:
    """Sequences the reference's `Trainer.train` over an engine-backed model.

    make_loaders(task, kind, exemplars) -> {group: loader} for kind in ('train', 'val', 'memory'); a loader is iterable over
        batches (dicts for `train_step`) and has `len(loader.dataset)` like a DataLoader.  'memory' receives the exemplar list.
    task_items(task) -> list of the task's training question dicts (each with 'img_id'); only needed when memory is on.
    make_optimizer(model, lr) -> optimizer (default: FusedAdamW over the reference's parameter groups, clip fused).
    Hooks: evaluate(loader) after every epoch, save(name) / test(task) after every task; `on_event(kind, **info)` observes
    the schedule (used by the tests to compare the order of operations).
    """

    def __init__(self, model, make_loaders, task_items=None, img_cate_map=None, *, task_list=ALL_TASKS, groups=CATEGORY_SPLITS,
                 epochs=3, batch_size=80, lr=1e-4, warmup_ratio=0.05, weight_decay=0.01, adam_eps=1e-6, clip_grad_norm=5.0,
                 proto_alpha=0.5, proto_beta=0.3, m_size=5000, memory=True, comp_cate="G-1", make_optimizer=None, evaluate=None,
                 save=None, test=None, on_event=None, rng=_random, barrier=None):
        self.model, self.make_loaders, self.task_items, self.img_cate_map = model, make_loaders, task_items, img_cate_map
        self.task_list, self.groups = list(task_list), groups
        self.epochs, self.batch_size, self.lr, self.warmup_ratio = epochs, batch_size, lr, warmup_ratio
        self.weight_decay, self.adam_eps, self.clip_grad_norm = weight_decay, adam_eps, clip_grad_norm
        self.proto_alpha, self.proto_beta = proto_alpha, proto_beta
        self.use_memory, self.comp_cate = memory, comp_cate
        self.memory = ExemplarMemory(m_size, groups)
        self.make_optimizer = make_optimizer or self._default_optimizer
        self.evaluate, self.save, self.test = evaluate, save, test
        self.on_event = on_event or (lambda kind, **info: None)
        self.rng, self.barrier = rng, barrier or (lambda: None)
        self.task_iftrain = {t: 0 for t in self.task_list}
        self.task_total_num = torch.zeros(len(self.task_list))
        self.optim = self.lr_scheduler = None

    def _default_optimizer(self, model, lr):
        from .optim import FusedAdamW, reference_param_groups
        return FusedAdamW(reference_param_groups(model, self.weight_decay), model, lr=lr, eps=self.adam_eps,
                          max_grad_norm=self.clip_grad_norm if self.clip_grad_norm > 0 else None)

    def train_step(self, batch, task_idx, each_memory):
        """vqacl.py:429-509: forward, backward, clip (fused into the optimizer), optimizer step, schedule step, grads dropped."""
        results = self.model.train_step(batch, task_idx, self.proto_alpha, self.proto_beta, each_memory, self.task_total_num)
        results["loss"].backward()
        self.optim.step()
        if self.lr_scheduler:
            self.lr_scheduler.step()
        for p in self.model.parameters():
            p.grad = None
        lr = self.lr_scheduler.get_last_lr()[0] if self.lr_scheduler else self.lr
        return results, lr

    def train(self, start_task=0):
        """`start_task` > 0 resumes after a checkpoint of task start_task-1 (the reference's resume path indexes the memory with
        the restarted counter, vqacl.py:163-167; here the global task index is used throughout)."""
        for t in self.task_list[:start_task]:
            self.task_iftrain[t] = 1
        for task_idx in range(start_task, len(self.task_list)):
            task = self.task_list[task_idx]
            self.task_iftrain[task] = 1
            self.on_event("task", task=task, task_idx=task_idx)
            exemplars, each_memory = [], 0
            if self.use_memory and task_idx != start_task:
                exemplars, each_memory = self.memory.update(task_idx, self.task_items(self.task_list[task_idx - 1]),
                                                            self.img_cate_map, self.rng)
                self.on_event("memory", size=len(exemplars), each_memory=each_memory)
            train_loaders = self.make_loaders(task, "train", [])
            self.task_total_num[task_idx] = sum(len(l.dataset) for l in train_loaders.values())
            val_loaders = self.make_loaders(task, "val", [])
            memory_loaders = self.make_loaders(task, "memory", exemplars)
            self.barrier()
            for group in shuffled_groups(self.groups, self.rng):
                train_l, val_l, mem_l = train_loaders[group], val_loaders[group], memory_loaders[group]
                n_mem = len(mem_l.dataset)
                total = 2 * len(train_l.dataset) if n_mem > 0 else len(train_l.dataset)
                self.optim = self.make_optimizer(self.model, self.lr)
                n_warm = warmup_iters(total, self.batch_size, self.epochs, self.warmup_ratio)
                self.lr_scheduler = constant_schedule_with_warmup(self.optim, n_warm)
                self.on_event("group", task=task, group=group, total_train_num=total, warmup_iters=n_warm)
                if group == self.comp_cate and task_idx != start_task:
                    self.on_event("skip", task=task, group=group)
                    continue
                for epoch in range(self.epochs):
                    self.model.train()
                    if hasattr(getattr(train_l, "sampler", None), "set_epoch"):
                        train_l.sampler.set_epoch(epoch)
                    pairs = zip(train_l, cycle(mem_l)) if n_mem > 0 else ((b, None) for b in train_l)
                    loss = loss_mem = None
                    for batch, mem_batch in pairs:
                        results, lr = self.train_step(batch, task_idx, each_memory)
                        loss = results["loss"]
                        self.on_event("step", source="new", task_idx=task_idx, lr=lr)
                        if mem_batch:
                            results_mem, lr = self.train_step(mem_batch, task_idx, each_memory)
                            loss_mem = results_mem["loss"]
                            self.on_event("step", source="memory", task_idx=task_idx, lr=lr)
                        self.barrier()
                    self.on_event("epoch", task=task, group=group, epoch=epoch,
                                  loss=None if loss is None else float(loss.detach()),
                                  loss_mem=None if loss_mem is None else float(loss_mem.detach()))
                    if self.evaluate:
                        self.evaluate(val_l)
                    self.barrier()
            if self.save:
                self.save(task + "_LAST")
            self.on_event("saved", name=task + "_LAST")
            if self.test:
                self.test(task)
