"""Leave-one-out evaluator: the caller of the scoring path (``Code/Recommender/evaluate.py``).

``evaluate_model`` keeps the reference's name, argument order and return value
(``evaluate.py:13-32``) but scores and ranks every test user in ONE device launch
(``m2d_rank_candidates``) instead of one ``sess.run`` per user (``evaluate.py:28-31, :58``).
Per-user results are identical, including the two quirks of ``eval_one_rating``:

* a dish that occurs twice among the candidates keeps its first position and its last score
  (dict insertion, ``evaluate.py:60-61``);
* ``heapq.nlargest`` breaks score ties toward the earlier candidate (``evaluate.py:63``).

A user whose candidates include a NaN score (a dish with an all-zero category mask, 0/0 at
``Model_Recommender.py:79``) is re-ranked on the host with the reference's own dict + ``heapq``
sequence, since Python's ordering of NaN keys is whatever that sequence does.
"""
from __future__ import annotations

import heapq
import itertools
import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from .recommender import Model


def getHitRatio(ranklist: Sequence, gtItem) -> int:
    """1 if the held-out dish is in the top-K list (evaluate.py:69-73)."""
    return 1 if any(item == gtItem for item in ranklist) else 0


def getNDCG(ranklist: Sequence, gtItem) -> float:
    """ln 2 / ln(rank + 2) at the held-out dish's 0-based rank, else 0 (evaluate.py:76-81)."""
    for rank, item in enumerate(ranklist):
        if item == gtItem:
            return math.log(2) / math.log(rank + 2)
    return 0


def _candidates(user, testRatings, testNegatives) -> List[int]:
    # evaluate.py:39-51: the positive first, then negatives 50..99 of that user
    return [testRatings[str(user)][0]] + list(testNegatives[str(user)][50:100])


# evaluate.py:5-10: the reference's evaluator keeps its arguments in module globals, set by evaluate_model and read by
# eval_one_rating(user).  Kept for the one-argument call; the six-argument form needs none of them.
_model = None
_testRatings = None
_testNegatives = None
_K = None
_dish_to_category = None


def eval_one_rating(*args):
    """Reference-shaped single-user path (evaluate.py:35-66): one predict call of <= 51 pairs, host ranking.

    ``eval_one_rating(user)`` -- the reference's own signature (evaluate.py:35): the model and the split are those of the
    last ``evaluate_model`` call, as in the reference's module globals (evaluate.py:5-10, :14-25).
    ``eval_one_rating(model, user, testRatings, testNegatives, K, dish_to_category)`` -- the same with everything passed in.
    Returns ``(hit, ndcg)``, or ``None`` for a user without a test rating (evaluate.py:37-38)."""
    if len(args) == 1:
        if _model is None:
            raise RuntimeError("eval_one_rating(user): call evaluate_model first (it sets the evaluator's model and split, "
                               "evaluate.py:14-25), or pass (model, user, testRatings, testNegatives, K, dish_to_category)")
        model, user = _model, args[0]
        testRatings, testNegatives, K, dish_to_category = _testRatings, _testNegatives, _K, _dish_to_category
    elif len(args) == 6:
        model, user, testRatings, testNegatives, K, dish_to_category = args
    else:
        raise TypeError("eval_one_rating takes (user) or (model, user, testRatings, testNegatives, K, dish_to_category)")
    if str(user) not in testRatings or len(testRatings[str(user)]) == 0:
        return None
    items = _candidates(user, testRatings, testNegatives)
    cats = [dish_to_category[str(i)] for i in items]
    scores = model.predict([user] * len(items), items, cats)
    table = {}
    for it, sc in zip(items, scores):
        table[it] = sc
    ranklist = heapq.nlargest(K, table, key=table.get)
    return getHitRatio(ranklist, items[0]), getNDCG(ranklist, items[0])


class _EvalPlan:
    """Everything `evaluate_model` derives from its three dict arguments, built once and kept on the device: the flat
    candidate array (evaluate.py:39-51 for every user), its lengths, the user ids and the dish -> category table."""

    __slots__ = ("refs", "held", "stamp", "users", "users_dev", "items_dev", "lens_dev", "gt", "dish_table", "engine_key")


_PLANS: "List[_EvalPlan]" = []          # most recent first; a training run alternates between at most a few splits
_MAX_PLANS = 4


def clear_eval_plans():
    """Forget the cached evaluation plans (call after editing ELEMENTS of a split's lists in place)."""
    del _PLANS[:]


def _held(testRatings, testNegatives, dish_to_category):
    """The users in order and every dict's value objects in order.  Kept by the plan (so no replaced list can be freed
    and its id recycled) and compared with `==` on the next call: list comparison checks identity first, so an unchanged
    entry costs a pointer compare and a replaced list is compared by content -- no replaced list can be missed."""
    return (list(testRatings), list(testRatings.values()), len(testNegatives), list(testNegatives.values()),
            len(dish_to_category), list(dish_to_category.values()))


def _stamp(testRatings, testNegatives, dish_to_category):
    """What identity cannot see -- an element appended to, removed from or rewritten in a list that is still the same
    object: every list's length, plus the contents of up to 64 evenly spaced users' candidate lists.  An element-wise
    in-place edit that keeps the length and misses the sample needs `clear_eval_plans()`; the driver never edits its
    split between epochs (Train_recommender.py:124-133 load it once)."""
    keys = list(testRatings.keys())
    step = max(1, len(keys) // 64)
    probe = tuple((k, tuple(testRatings[k][:1]), tuple(testNegatives[k][50:100]) if k in testNegatives else None)
                  for k in keys[::step][:64])
    return (sum(map(len, testRatings.values())), sum(map(len, testNegatives.values())),
            sum(map(len, dish_to_category.values())), probe)


def _engine_key(engine):
    # a serial that is never reused (id() of a collected engine can be) plus what a plan's tensors depend on
    return engine.id, engine.I, engine.C, str(engine.device)


def _build_plan(model: Model, testRatings, testNegatives, dish_to_category) -> _EvalPlan:
    users = list(testRatings.keys())
    for u in users:
        if len(testRatings[u]) == 0:
            # the reference's eval_one_rating returns None here and the caller's tuple-unpack fails
            raise TypeError("cannot unpack non-iterable NoneType object (user %s has no test rating)" % u)
    # evaluate.py:39-51 for every user at once: the positive first, then negatives 50..99.  One flat int array;
    # the per-user Python work is one list concatenation.
    cand = [_candidates(u, testRatings, testNegatives) for u in users]
    lens = np.fromiter(map(len, cand), dtype=np.int32, count=len(cand))
    flat = np.fromiter(itertools.chain.from_iterable(cand), dtype=np.int64, count=int(lens.sum()))
    for it in np.unique(flat).tolist():
        if str(it) not in dish_to_category:
            raise KeyError(str(it))                          # evaluate.py:43 / :50 would raise the same
    L = int(lens.max())
    if L > 1024:
        raise ValueError("evaluate_model: at most 1024 candidates per user on the device path")
    if flat.size and (flat.min() < -(2 ** 31) or flat.max() >= 2 ** 31):
        raise IndexError("item id does not fit int32")
    if int(lens.min()) == L:
        items_np = flat.astype(np.int32).reshape(len(cand), L)
    else:
        items_np = np.zeros((len(cand), L), dtype=np.int32)
        items_np[np.arange(L)[None, :] < lens[:, None]] = flat
    users_np = np.fromiter(map(int, users), dtype=np.int64, count=len(users))
    if users_np.min() < -(2 ** 31) or users_np.max() >= 2 ** 31:
        raise IndexError("user id does not fit int32")
    dev = model.engine.device
    p = _EvalPlan()
    p.refs = (testRatings, testNegatives, dish_to_category)      # strong: keeps the ids in the cache key from being recycled
    p.held = _held(testRatings, testNegatives, dish_to_category)
    p.stamp = _stamp(testRatings, testNegatives, dish_to_category)
    p.users = users
    p.users_dev = torch.from_numpy(users_np.astype(np.int32)).to(dev)
    p.items_dev = torch.from_numpy(items_np).to(dev)
    p.lens_dev = torch.from_numpy(lens).to(dev)
    p.gt = items_np[:, 0].copy()
    model.set_dish_categories(dish_to_category)               # dict -> [I, C] table, copied to HBM once
    p.dish_table = model.engine.dish_cats                     # the resident tensor itself (kept alive by the plan)
    p.engine_key = _engine_key(model.engine)
    return p


def _plan_for(model: Model, testRatings, testNegatives, dish_to_category) -> _EvalPlan:
    key = _engine_key(model.engine)
    for i, p in enumerate(_PLANS):
        if (p.refs[0] is testRatings and p.refs[1] is testNegatives and p.refs[2] is dish_to_category
                and p.engine_key == key and p.held == _held(testRatings, testNegatives, dish_to_category)
                and p.stamp == _stamp(testRatings, testNegatives, dish_to_category)):
            if i:
                _PLANS.insert(0, _PLANS.pop(i))
            return p
    p = _build_plan(model, testRatings, testNegatives, dish_to_category)
    _PLANS.insert(0, p)
    del _PLANS[_MAX_PLANS:]
    return p


def evaluate_model(sess, model: Model, testRatings: Dict[str, List[int]], testNegatives: Dict[str, List[int]],
                   K: int, dish_to_category: Dict[str, list]) -> Tuple[List[int], List[float]]:
    """HR@K / NDCG@K for every user of ``testRatings``, in dict order (evaluate.py:13-32).

    ``sess`` is accepted for signature parity (a ``foodrec_amd.Session`` or ``None``); the launch goes
    through ``model.engine``.  The driver calls this every ``verbose`` epochs with the same three dicts
    (Train_recommender.py:210): the candidate arrays and the dish table built from them are kept on the device
    (``_EvalPlan``), so a repeat call costs one launch, one device-to-host copy of ``[users, K]`` ids, the HR / NDCG
    arithmetic and the check that the plan still describes its dicts -- ``_held`` / ``_stamp``: O(users + dishes) on the
    host (lists of the three dicts' values compared by identity, every list's length summed; a few ms at the reference's
    64 657 users and 4 548 dishes, and it grows with a ``dish_to_category`` of millions of dishes), but none of the
    candidate building of the first call.  The plan is reused only while the three
    dicts are the same objects holding the same keys and the same (or equal) list objects of the same lengths; rewriting
    elements of a list in place calls for ``clear_eval_plans()``.

    Like the reference, which feeds ``categories`` from ``dish_to_category`` on every call (evaluate.py:43, :50), this
    makes ``dish_to_category`` the engine's resident dish mask table: a table set earlier with ``set_dish_categories`` is
    replaced.
    """
    global _model, _testRatings, _testNegatives, _K, _dish_to_category
    _model, _testRatings, _testNegatives, _K, _dish_to_category = model, testRatings, testNegatives, K, dish_to_category
    if not testRatings:
        return [], []
    if K > 64:
        raise ValueError("evaluate_model: K <= 64 on the device path")
    plan = _plan_for(model, testRatings, testNegatives, dish_to_category)
    eng = model.engine
    if eng.dish_cats is None or eng.dish_cats.data_ptr() != plan.dish_table.data_ptr():
        eng.set_dish_categories(plan.dish_table)             # another mask table is resident: put ours back (a pointer, no copy)
    s, ids, flags = eng.rank_candidates(plan.users_dev, plan.items_dev, int(K), lens=plan.lens_dev)
    eng.check()
    ids = ids.cpu().numpy()
    flags = flags.cpu().numpy()

    # getHitRatio / getNDCG (evaluate.py:69-81) for all users: 0-based rank of the held-out dish in its list.
    # The list holds each dish once (dict collapse) and padding is -1, so the first match is the only one.
    match = ids == plan.gt[:, None]
    hit = match.any(axis=1)
    rank = match.argmax(axis=1)
    gain = np.array([math.log(2) / math.log(r + 2) for r in range(ids.shape[1])])     # the reference's own expression
    hits: List[int] = hit.astype(np.int64).tolist()
    ndcgs: List[float] = np.where(hit, gain[rank], 0.0).tolist()
    for r in np.flatnonzero(flags & 1).tolist():            # NaN among the scores: the reference's host sequence
        hits[r], ndcgs[r] = eval_one_rating(model, plan.users[r], testRatings, testNegatives, K, dish_to_category)
    return hits, ndcgs
