"""VQA answer normalisation, soft accuracy and the continual-learning metrics.

Reference: VL-T5/src/vqa_data_memory.py:983-1199 (`VQAEvaluator`, itself the official VQA evaluation procedure: punctuation and
digit/article normalisation, accuracy = min(1, #matching human answers / 3) averaged over the ten leave-one-out subsets) and
Question_type.py:107-201 (`evaluate_metric`: incremental average accuracy and forgetting from the task x task result matrix).

Host-side string and table work: nothing here touches the GPU.  Results are compared with the reference's own evaluator run
on the same inputs (tests/golden/g8_evaluator.json).
"""
import json
import re

from .loop import ALL_TASKS, COMP_TASKS

# The evaluation's contraction list in canonical spelling.  The lookup table maps every spelling with ONE apostrophe dropped to
# the canonical form ("couldnt've", "couldn'tve" -> "couldn't've").
_CONTRACTIONS = """'ow's'at 'twas ain't aren't can't could've couldn't couldn't've didn't doesn't don't hadn't hadn't've hasn't
haven't he'd he'd've he's how'd how'll how's I'd've I'm I've isn't it'd it'd've it'll ma'am might've mightn't mightn't've must've
mustn't needn't not've o'clock oughtn't shan't she'd've should've shouldn't shouldn't've somebody'd've somebody'll somebody's
someone'd someone'd've someone'll someone's something'd something'd've something'll that's there'd there'd've there're there's
they'd they'd've they'll they're they've wasn't we'd've we've weren't what'll what're what's what've when's where'd where's
where've who'd who'd've who'll who's who've why'll why're why's won't would've wouldn't wouldn't've y'all y'all'd've y'all'll
you'd you'd've you'll you're you've""".split()


def _contraction_table():
    table = {}
    for word in _CONTRACTIONS:
        for i, ch in enumerate(word):
            if ch == "'":
                table[word[:i] + word[i + 1:]] = word
    # the three entries of the published table that do not follow the rule (two identities and one reversed pair)
    table["let's"] = "let's"
    table["she's"] = "she's"
    table["somebody'd"] = "somebodyd"
    return table


CONTRACTIONS = _contraction_table()
NUMBER_WORDS = {"none": "0", "zero": "0", "one": "1", "two": "2", "three": "3", "four": "4", "five": "5", "six": "6", "seven": "7",
                "eight": "8", "nine": "9", "ten": "10"}
ARTICLES = ("a", "an", "the")
PUNCTUATION = [";", "/", "[", "]", '"', "{", "}", "(", ")", "=", "+", "\\", "_", "-", ">", "<", "@", "`", ",", "?", "!"]
_PERIOD = re.compile(r"(?!<=\d)(\.)(?!\d)")       # as published (the look-behind is spelled `(?!<=`): a period not followed by a digit
_COMMA_IN_NUMBER = re.compile(r"(\d)(\,)(\d)")


def process_punctuation(text):
    """Drop a punctuation mark when it touches a space anywhere in the answer (or the answer holds a digit,digit comma), else turn
    it into a space; then remove periods that are not followed by a digit (vqa_data_memory.py:1156-1166)."""
    out = text
    digits_comma = _COMMA_IN_NUMBER.search(text) is not None
    for p in PUNCTUATION:
        if (p + " " in text or " " + p in text) or digits_comma:
            out = out.replace(p, "")
        else:
            out = out.replace(p, " ")
    # the reference passes re.UNICODE (= 32) in the position of `count`: at most 32 periods are removed
    return _PERIOD.sub("", out, 32)


def process_digit_article(text):
    """Lower-case, number words -> digits, articles removed, contractions restored (vqa_data_memory.py:1168-1181)."""
    words = []
    for w in text.lower().split():
        w = NUMBER_WORDS.get(w, w)
        if w not in ARTICLES:
            words.append(w)
    return " ".join(CONTRACTIONS.get(w, w) for w in words)


def _clean(ans):
    ans = ans.replace("\n", " ").replace("\t", " ").strip()
    return process_digit_article(process_punctuation(ans))


def normalize_answer(ans):
    """`VQAEvaluator.normalize_answer` (:1147-1154): the target-side normaliser (`--answer_normalize`)."""
    return _clean(ans).replace(",", "")


class VQAEvaluator:
    """`dataset` carries `id2datum` (question_id -> {'label': {answer: score}, optional 'is_topk_optimal'}) and `id2datum_gt`
    (question_id -> official annotation: 'answers' [{'answer'}...], 'question_type', 'answer_type') like the reference's
    `VQADataset` (:914-981)."""

    def __init__(self, dataset=None):
        self.dataset = dataset
        self.n = 2

    def evaluate(self, quesid2ans):
        """Top-k soft score: mean over questions of label[answer] (0 when the answer is not a label key) (:1039-1046)."""
        score = 0.0
        for qid, ans in quesid2ans.items():
            label = self.dataset.id2datum[qid]["label"]
            if ans in label:
                score += label[ans]
        return score / len(quesid2ans)

    def dump_result(self, quesid2ans, path):
        with open(path, "w") as f:
            json.dump([{"question_id": q, "answer": a} for q, a in quesid2ans.items()], f, indent=4, sort_keys=True)

    def evaluate_raw(self, quesid2ans, is_topk_optimal=None):
        """Official VQA accuracy in percent: overall, per question type, per answer type (:1069-1145).  Like the reference, the
        human answers of a question are punctuation-normalised IN PLACE (in `id2datum_gt`) when they are not all identical."""
        gts = self.dataset.id2datum_gt
        self.accuracy, self.evalQA, self.evalQuesType, self.evalAnsType = {}, {}, {}, {}
        acc_all, acc_qtype, acc_atype = [], {}, {}
        for qid, res in quesid2ans.items():
            qid = int(qid)
            datum = self.dataset.id2datum[qid]
            if is_topk_optimal is not None and "is_topk_optimal" in datum and datum["is_topk_optimal"] != is_topk_optimal:
                continue
            res = _clean(res)
            humans = gts[qid]["answers"]
            if len(set(a["answer"] for a in humans)) > 1:
                for a in humans:
                    a["answer"] = process_punctuation(a["answer"])
            accs = []
            for held_out in humans:
                # `!=` on the annotation dicts, as the reference: every annotation EQUAL to the held-out one is left out too
                matching = [a for a in humans if a != held_out and a["answer"] == res]
                accs.append(min(1, float(len(matching)) / 3))
            acc = float(sum(accs)) / len(accs)
            qtype, atype = gts[qid]["question_type"], gts[qid]["answer_type"]
            acc_all.append(acc)
            acc_qtype.setdefault(qtype, []).append(acc)
            acc_atype.setdefault(atype, []).append(acc)
            self.evalQA[qid] = round(100 * acc, self.n)
            self.evalQuesType.setdefault(qtype, {})[qid] = round(100 * acc, self.n)
            self.evalAnsType.setdefault(atype, {})[qid] = round(100 * acc, self.n)
        if not acc_all:
            return {"overall": 0, "perQuestionType": {}, "perAnswerType": {}}
        self.accuracy["overall"] = round(100 * float(sum(acc_all)) / len(acc_all), self.n)
        self.accuracy["perQuestionType"] = {k: round(100 * float(sum(v)) / len(v), self.n) for k, v in acc_qtype.items()}
        self.accuracy["perAnswerType"] = {k: round(100 * float(sum(v)) / len(v), self.n) for k, v in acc_atype.items()}
        return self.accuracy

    def normalize_answer(self, ans):
        return normalize_answer(ans)


def result_matrix(results, start=0):
    """results[trained_task][tested_task] = accuracy -> lower-triangular matrix (rows: after training task i), -1 where a task
    was not yet seen (Question_type.py:108-117)."""
    keys = list(results)
    n = len(keys)
    m = [[-1.0] * n for _ in range(n)]
    for i in range(start, n):
        for j in range(start, i + 1):
            m[i][j] = results[keys[i]][keys[j]]
    return m


def evaluate_metric(results, start=0, all_tasks=ALL_TASKS, comp_tasks=COMP_TASKS):
    """Incremental average accuracy / forgetting, over all tasks and over the 6 composition tasks (Question_type.py:107-201).
    Forgetting of task j after task t = (best accuracy on j after any earlier task) - (accuracy on j after t)."""
    comp_idx = [all_tasks.index(t) for t in comp_tasks]
    m = result_matrix(results, start)
    n = len(m)
    inc_acc, inc_acc_6q = [], []
    for t in range(start, n):
        seen = [a for a in m[t] if a != -1]
        inc_acc.append(sum(seen) / len(seen))
        seen6 = [m[t][i] for i in range(n) if i in comp_idx and m[t][i] != -1]
        inc_acc_6q.append(sum(seen6) / len(seen6) if seen6 else -1)
    inc_forget, inc_forget_6q = [0], [0]
    for t in range(1 + start, n):
        forget = []
        for j in range(start, t):
            best_before = max(m[i][j] for i in range(t))
            forget.append(0 if best_before == -1 else best_before - m[t][j])
        inc_forget.append(sum(forget) / len(forget))
        # the reference selects entry i_ of the forgetting list when i_+1 is a composition-task index (:177-179)
        forget6 = [forget[i] for i in range(len(forget)) if i + 1 in comp_idx]
        inc_forget_6q.append(sum(forget6) / len(forget6) if forget6 else -1)
    return {"Incre_avg_acc": inc_acc, "Avg_acc": inc_acc[-1], "Incre_avg_forget": inc_forget, "Avg_forget": inc_forget[-1],
            "Incre_avg_acc_6Q": inc_acc_6q, "Avg_acc_6Q": inc_acc_6q[-1], "Incre_avg_forget_6Q": inc_forget_6q,
            "Avg_forget_6Q": inc_forget_6q[-1]}
