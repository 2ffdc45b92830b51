"""Generate evaluator / file-format golden vectors by RUNNING THE REFERENCE'S OWN pure-Python modules.

Runs only in the build container (needs /root/reference); the fixtures it writes are committed and
are what travels.  What is executed from the reference:

* ``Code/Recommender/evaluate.py`` (imports only ``math`` and ``heapq``): ``evaluate_model`` is driven
  with a duck-typed ``sess``/``model`` pair -- the two arguments the function takes -- whose ``run``
  returns scores from a table stored in the fixture.  This pins rows A8/A9 (candidate batch
  ``[positive] + negatives[50:100]``, dict collapse, ``heapq.nlargest`` tie order, HR/NDCG).
* ``Code/Recommender/Dataset.py`` (no imports): its three readers parse files written by this script;
  this pins the on-disk formats.

The TensorFlow part of the reference (Model_Recommender.py) is NOT run -- TF is absent -- so nothing
here pins the scoring arithmetic.

Usage:  python tests/golden/make_reference_eval_golden.py
"""
import importlib.util
import json
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference/Code/Recommender"
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Model:
    """Placeholder handles used as feed_dict keys by evaluate.py:55-57."""
    user_input, item_input, labels, categories = "user_input", "item_input", "labels", "categories"
    dropout_keep_prob, is_training_flag, logits = "dropout_keep_prob", "is_training_flag", "logits"


class _ModelSess:
    """`sess.run([model.logits], feed_dict)` -> [scores] computed by the build's float32 restatement of
    Model.inference (oracle/m2d_oracle.py) on tables whose values are small dyadic rationals, so every
    product and partial sum is exact in float32 and the scores do not depend on summation order."""

    def __init__(self, PM, RE, CE, coef=0.99):
        self.PM, self.RE, self.CE, self.coef = PM, RE, CE, coef

    def run(self, fetches, feed_dict):
        assert fetches == [_Model.logits]
        from oracle import m2d_oracle
        return [m2d_oracle.inference_f32(self.PM, self.RE, self.CE, feed_dict[_Model.user_input],
                                         feed_dict[_Model.item_input], feed_dict[_Model.categories], self.coef)]


class _Sess:
    """`sess.run([model.logits], feed_dict)` -> [scores]; scores come from a dense table S[user, item]."""

    def __init__(self, table):
        self.table = table
        self.calls = []

    def run(self, fetches, feed_dict):
        assert fetches == [_Model.logits]
        users = [int(u) for u in feed_dict[_Model.user_input]]      # ids arrive as str (evaluate.py:28, :41)
        items = [int(i) for i in feed_dict[_Model.item_input]]
        self.calls.append({"n": len(users), "cats": np.asarray(feed_dict[_Model.categories]).shape})
        return [self.table[users, items].astype(np.float32)]


def make_eval_case(seed, n_users, n_items, K, mode):
    rng = np.random.default_rng(seed)
    cats = {str(d): [[float(x)] for x in rng.integers(0, 2, 4)] for d in range(n_items)}
    ratings, negatives = {}, {}
    for u in range(n_users):
        if mode == "short" and u % 3 == 0:
            nneg = 50 + int(rng.integers(0, 30))       # fewer than 100 negatives -> short candidate list
        else:
            nneg = 100 + int(rng.integers(0, 5))
        pos = int(rng.integers(0, n_items))
        neg = [int(x) for x in rng.integers(0, n_items, nneg)]   # with replacement: duplicates, may hit pos
        ratings[str(u)] = [pos, int(rng.integers(0, n_items))]
        negatives[str(u)] = neg
    if mode == "ties":
        table = rng.integers(0, 4, (n_users, n_items)).astype(np.float32) * 0.25   # heavy ties
    else:
        table = rng.standard_normal((n_users, n_items)).astype(np.float32)
    return ratings, negatives, cats, table, K


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present; fixtures are generated in the build container only")
    ev = _load("evaluate")
    ds = _load("Dataset")

    # ---- evaluator -----------------------------------------------------------------------------
    cases = []
    for seed, nu, ni, K, mode in [(1, 5, 40, 10, "ties"), (2, 7, 300, 10, "random"), (3, 6, 30, 3, "ties"),
                                  (4, 9, 60, 10, "short"), (5, 4, 8, 10, "ties")]:
        ratings, negatives, cats, table, K = make_eval_case(seed, nu, ni, K, mode)
        sess = _Sess(table)
        hits, ndcgs = ev.evaluate_model(sess, _Model, ratings, negatives, K, cats)
        # per-user rank lists straight from the reference's own nlargest call
        ranklists = []
        for u in ratings:
            items = [ratings[u][0]] + negatives[u][50:100]
            scores = table[int(u), items]
            d = {}
            for it, sc in zip(items, scores):
                d[it] = sc
            import heapq
            ranklists.append([int(x) for x in heapq.nlargest(K, d, key=d.get)])
        cases.append({"seed": seed, "mode": mode, "K": K, "testRatings": ratings,
                      "testNegatives": negatives, "dish_to_category": cats,
                      "score_table": [[float(x) for x in row] for row in table],
                      "hits": [int(h) for h in hits], "ndcgs": [float(x) for x in ndcgs],
                      "ranklists": ranklists, "calls": [c["n"] for c in sess.calls]})
    with open(os.path.join(OUT, "ref_eval_cases.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_reference_eval_golden.py",
                   "reference": "Code/Recommender/evaluate.py:13-81 executed as-is", "cases": cases}, f)

    # ---- evaluator driven by model scores (device-path parity: tests/test_gpu_evaluator.py) ------
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    mcases = []
    # the last four (round 5): other values of --high_level_score_coefficient (Train_recommender.py:61-62) than its default
    # 0.99 -- with 1.0 every dish of a mask pattern scores the same, so nlargest's tie order decides most of the lists
    for seed, nu, ni, E, K, n_nan, coef in [(11, 9, 40, 8, 10, 0, 0.99), (12, 6, 25, 32, 10, 1, 0.99), (13, 12, 400, 64, 5, 3, 0.99),
                                           (14, 8, 60, 200, 10, 0, 0.99), (15, 10, 50, 64, 10, 1, 0.5), (16, 8, 40, 32, 10, 0, 1.0),
                                           (17, 7, 60, 64, 5, 2, 0.0), (18, 9, 45, 128, 10, 0, 1.25)]:
        rng = np.random.default_rng(seed)
        q = lambda shape: (rng.integers(-4, 5, shape) / 4.0).astype(np.float32)     # dyadic values
        PM, RE, CE = q((nu, 5, E)), q((ni, E)), q((4, E))
        RE[ni // 2:ni // 2 + 5] = RE[0:5]                   # identical dishes -> exact score ties
        pat = rng.integers(1, 16, ni)
        if n_nan:
            pat[rng.integers(5, ni // 2, n_nan)] = 0         # dishes with an empty mask -> NaN scores
        catm = ((pat[:, None] >> np.arange(4)[None, :]) & 1).astype(np.float32)
        catm[ni // 2:ni // 2 + 5] = catm[0:5]
        cats = {str(d): [[float(x)] for x in catm[d]] for d in range(ni)}
        ratings, negatives = {}, {}
        for u in range(nu):
            ratings[str(u)] = [int(rng.integers(0, ni))]
            nneg = 100 if u % 4 else 60 + int(rng.integers(0, 30))
            negatives[str(u)] = [int(x) for x in rng.integers(0, ni, nneg)]
        hits, ndcgs = ev.evaluate_model(_ModelSess(PM, RE, CE, coef), _Model, ratings, negatives, K, cats)
        mcases.append({"seed": seed, "K": K, "E": E, "coef": coef, "PM": PM.tolist(), "RE": RE.tolist(), "CE": CE.tolist(),
                       "testRatings": ratings, "testNegatives": negatives, "dish_to_category": cats,
                       "hits": [int(h) for h in hits], "ndcgs": [float(x) for x in ndcgs]})
    with open(os.path.join(OUT, "ref_eval_model_cases.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_reference_eval_golden.py",
                   "reference": "Code/Recommender/evaluate.py:13-81 executed as-is; scores from the build's "
                                "float32 restatement of Model.inference (TF unavailable)", "cases": mcases}, f)

    # ---- file formats ---------------------------------------------------------------------------
    rng = np.random.default_rng(77)
    train_lines, test_lines, neg_lines = [], [], []
    for u in range(6):
        for _ in range(int(rng.integers(1, 5))):
            train_lines.append("%d\t%d\t%d\t%d\n" % (u, rng.integers(0, 50), rng.integers(1, 6), rng.integers(1e6)))
        test_lines.append("%d\t%d\n" % (u, rng.integers(0, 50)))
        negs = "\t".join(str(int(x)) for x in rng.integers(0, 50, 100))
        neg_lines.append("(%d)\t%s\n" % (u, negs))
    with tempfile.TemporaryDirectory() as td:
        base = os.path.join(td, "toy")
        for suf, lines in ((".train.rating", train_lines), (".test.rating", test_lines),
                           (".test.negative", neg_lines)):
            with open(base + suf, "w") as f:
                f.writelines(lines)
        d = ds.Dataset(base)
        fmt = {"files": {".train.rating": "".join(train_lines), ".test.rating": "".join(test_lines),
                         ".test.negative": "".join(neg_lines)},
               "trainMatrix": d.trainMatrix, "testRatings": d.testRatings, "testNegatives": d.testNegatives,
               "num_train_users": d.num_train_users, "num_instances": d.num_instances, "num_test": d.num_test}
    with open(os.path.join(OUT, "ref_dataset_format.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_reference_eval_golden.py",
                   "reference": "Code/Recommender/Dataset.py:3-71 executed as-is", "case": fmt}, f)
    print("wrote ref_eval_cases.json, ref_eval_model_cases.json, ref_dataset_format.json")


if __name__ == "__main__":
    main()
