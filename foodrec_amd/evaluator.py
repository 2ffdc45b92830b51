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


def eval_one_rating(model: Model, user, testRatings, testNegatives, K, dish_to_category):
    """Reference-shaped single-user path (evaluate.py:35-66): one predict call of <= 51 pairs, host ranking."""
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


def evaluate_model(sess, model: Model, testRatings: Dict[str, List[int]], testNegatives: Dict[str, List[int]],
                   K: int, dish_to_category: Dict[str, list]) -> Tuple[List[int], List[float]]:
    """HR@K / NDCG@K for every user of ``testRatings``, in dict order (evaluate.py:13-32).

    ``sess`` is accepted for signature parity (a ``foodrec_amd.Session`` or ``None``); the launch goes
    through ``model.engine``.
    """
    users = list(testRatings.keys())
    for u in users:
        if len(testRatings[u]) == 0:
            # the reference's eval_one_rating returns None here and the caller's tuple-unpack fails
            raise TypeError("cannot unpack non-iterable NoneType object (user %s has no test rating)" % u)
    if not users:
        return [], []
    # evaluate.py:39-51 for every user at once: the positive first, then negatives 50..99.  One flat int array;
    # the per-user Python work is one list concatenation.
    cand = [_candidates(u, testRatings, testNegatives) for u in users]
    lens = np.fromiter(map(len, cand), dtype=np.int32, count=len(cand))
    flat = np.fromiter(itertools.chain.from_iterable(cand), dtype=np.int64, count=int(lens.sum()))
    for it in np.unique(flat).tolist():
        if str(it) not in dish_to_category:
            raise KeyError(str(it))                          # evaluate.py:43 / :50 would raise the same
    L = int(lens.max())
    if L > 1024 or K > 64:
        raise ValueError("evaluate_model: at most 1024 candidates per user and K <= 64 on the device path")
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
    users_np = users_np.astype(np.int32)

    model.set_dish_categories(dish_to_category)
    eng = model.engine
    dev = eng.device
    s, ids, flags = eng.rank_candidates(torch.from_numpy(users_np).to(dev), torch.from_numpy(items_np).to(dev),
                                        int(K), lens=torch.from_numpy(lens).to(dev))
    eng.check()
    ids = ids.cpu().numpy()
    flags = flags.cpu().numpy()

    # getHitRatio / getNDCG (evaluate.py:69-81) for all users: 0-based rank of the held-out dish in its list.
    # The list holds each dish once (dict collapse) and padding is -1, so the first match is the only one.
    gt = items_np[:, 0]
    match = ids == gt[:, None]
    hit = match.any(axis=1)
    rank = match.argmax(axis=1)
    gain = [math.log(2) / math.log(r + 2) for r in range(ids.shape[1])]     # the reference's own expression
    hits: List[int] = hit.astype(np.int64).tolist()
    ndcgs: List[float] = [gain[r] if h else 0 for r, h in zip(rank.tolist(), hits)]
    for r in np.flatnonzero(flags & 1).tolist():            # NaN among the scores: the reference's host sequence
        hits[r], ndcgs[r] = eval_one_rating(model, users[r], testRatings, testNegatives, K, dish_to_category)
    return hits, ndcgs
