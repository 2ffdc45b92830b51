"""CPU oracle for the Market2Dish Recommender scoring path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
``foodrec_amd`` never imports anything under ``oracle/``.

PARITY STATUS -- read before trusting a number that came out of here:

* Rows A3-A7 (the arithmetic of ``Model.inference``,
  ``Code/Recommender/Model_Recommender.py:56-97``) execute inside TensorFlow 1.x,
  which is not in this image and cannot be installed (no network).  The reference
  ships no tests, no golden vectors and no fixtures for this path.  The functions
  below are therefore a *restatement* of those lines, op for op, checked against a
  hand-computed known answer (score = 3.4625) and against
  each other (float64 / float32-naive / factored / C).  **PARITY UNPINNED** for
  A3-A7: nothing produced by TensorFlow itself anchors these values.
* Rows A8-A9 (``Code/Recommender/evaluate.py:13-81``: candidate batch, dict
  collapse, ``heapq.nlargest``, HR/NDCG) are pure Python in the reference.  The
  restatement here is pinned against the reference's own module, imported from
  ``/root/reference`` in the build container by
  ``tests/golden/make_reference_eval_golden.py`` (fixtures:
  ``tests/golden/ref_eval_*.json``).

Every function cites the reference lines it follows (paths relative to
``/root/reference``).
"""
from __future__ import annotations

import heapq
import math
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

# Model_Recommender.py:17 -- tf.constant(args.high_level_score_coefficient) is a
# float32 constant; `1 - coef` (:96) is evaluated in float32 as well.
DEFAULT_COEF = 0.99


def blend_coefficients(coef: float = DEFAULT_COEF) -> Tuple[np.float32, np.float32]:
    """(a, 1-a) exactly as the graph holds them: both float32 (Model_Recommender.py:17, :95-96)."""
    a = np.float32(coef)
    return a, np.float32(np.float32(1.0) - a)


def _as_mask(categories, num_categories: int | None = None) -> np.ndarray:
    """Accept the placeholder layout [B, C, 1] (Model_Recommender.py:32) or [B, C]; return [B, C, 1]."""
    m = np.asarray(categories)
    if m.ndim == 2:
        m = m[:, :, None]
    if m.ndim != 3 or m.shape[2] != 1:
        raise ValueError("categories must be [B, C, 1] or [B, C], got %r" % (m.shape,))
    if num_categories is not None and m.shape[1] != num_categories:
        raise ValueError("categories has %d columns, model has %d" % (m.shape[1], num_categories))
    return m


def _as_ids(ids, bound: int, what: str) -> np.ndarray:
    """int32 feed conversion (Model_Recommender.py:26-29); TF-CPU GatherV2 rejects out-of-range ids."""
    a = np.asarray([int(x) for x in ids] if not isinstance(ids, np.ndarray) else ids)
    a = a.astype(np.int64).reshape(-1)
    if a.size and (a.min() < 0 or a.max() >= bound):
        bad = a[(a < 0) | (a >= bound)][0]
        raise IndexError("%s id %d out of range [0, %d)" % (what, int(bad), bound))
    return a


def inference(Personal_Memory, Recipe_Embedding, Category_Embedding, user_input, item_input,
              categories, coef: float = DEFAULT_COEF, dtype=np.float64) -> np.ndarray:
    """Op-for-op restatement of ``Model.inference`` (Model_Recommender.py:56-97).

    ``dtype=np.float32`` keeps every intermediate in float32 with the same
    materialised ``[B, C, E]`` temporaries the TF graph builds; ``np.float64``
    evaluates the same expression tree in double, with the two blend constants
    still rounded through float32 as the graph does (:17, :96).
    """
    PM = np.asarray(Personal_Memory)
    RE = np.asarray(Recipe_Embedding)
    CE = np.asarray(Category_Embedding)
    C = CE.shape[0]
    if PM.ndim != 3 or PM.shape[1] != C + 1:
        raise ValueError("Personal_Memory must be [U, C+1, E]")
    users = _as_ids(user_input, PM.shape[0], "user")
    items = _as_ids(item_input, RE.shape[0], "item")
    if users.shape != items.shape:
        raise ValueError("user_input and item_input differ in length")
    cat = _as_mask(categories, C).astype(dtype)                      # [B, C, 1]   :32
    if cat.shape[0] != users.shape[0]:
        raise ValueError("categories batch differs from ids")
    PM = PM.astype(dtype, copy=False)
    RE = RE.astype(dtype, copy=False)
    CE = CE.astype(dtype, copy=False)
    a32, b32 = blend_coefficients(coef)
    a, b = dtype(a32), dtype(b32)

    with np.errstate(divide="ignore", invalid="ignore"):
        User_Memory = PM[users]                                      # [B, C+1, E] :57
        U_high = User_Memory[:, :1, :]                               # [B, 1, E]   :59 split [1, C]
        U_low = User_Memory[:, 1:, :]                                # [B, C, E]
        Item = RE[items][:, None, :]                                 # [B, 1, E]   :63-65
        Dish_Category = cat * CE                                     # [B, C, E]   :67
        category_score = U_high * Dish_Category                      # [B, C, E]   :71
        sum_cat = category_score.sum(axis=(1, 2), dtype=dtype)       # [B]         :75
        category_num = cat.sum(axis=(1, 2), dtype=dtype)             # [B]         :77
        high_score = sum_cat / category_num                          # [B]         :79
        Dish_Memory = cat * U_low                                    # [B, C, E]   :82
        dish_score = Item * Dish_Memory                              # [B, C, E]   :86
        sum_dish = dish_score.sum(axis=(1, 2), dtype=dtype)          # [B]         :90
        low_score = sum_dish / category_num                          # [B]         :92
        score = a * high_score + b * low_score                       # [B]         :95-96
    return score.astype(dtype, copy=False)


def inference_f32(PM, RE, CE, users, items, categories, coef: float = DEFAULT_COEF) -> np.ndarray:
    return inference(PM, RE, CE, users, items, categories, coef, np.float32)


def inference_f64(PM, RE, CE, users, items, categories, coef: float = DEFAULT_COEF) -> np.ndarray:
    return inference(PM, RE, CE, users, items, categories, coef, np.float64)


def dish_vectors(Recipe_Embedding, Category_Embedding, dish_categories, coef: float = DEFAULT_COEF,
                 dtype=np.float64) -> np.ndarray:
    """Factored form of Model_Recommender.py:67-96: per dish d the vector

        Dt[d] = concat( a*(sum_c m_c CE_c)/n , (1-a)*m_0/n*RE[d], ..., (1-a)*m_{C-1}/n*RE[d] )

    so that score(u, d) = <flatten(PM[u]), Dt[d]>.  ``dish_categories`` is [I, C] (or [I, C, 1]).
    An independent second route to the same number, used to cross-check `inference`.
    """
    RE = np.asarray(Recipe_Embedding, dtype=dtype)
    CE = np.asarray(Category_Embedding, dtype=dtype)
    m = _as_mask(dish_categories, CE.shape[0])[:, :, 0].astype(dtype)    # [I, C]
    a32, b32 = blend_coefficients(coef)
    with np.errstate(divide="ignore", invalid="ignore"):
        n = m.sum(axis=1, dtype=dtype)                                   # [I]
        high = dtype(a32) * (m @ CE) / n[:, None]                        # [I, E]
        low = dtype(b32) * (m / n[:, None])[:, :, None] * RE[:, None, :]  # [I, C, E]
    return np.concatenate([high[:, None, :], low], axis=1).reshape(RE.shape[0], -1)


def inference_factored(PM, RE, CE, users, items, dish_categories, coef: float = DEFAULT_COEF,
                       dtype=np.float64) -> np.ndarray:
    """score = <flatten(PM[u]), Dt[d]> with Dt from `dish_vectors` (dish_categories indexed by dish id)."""
    PMf = np.asarray(PM, dtype=dtype)
    Dt = dish_vectors(RE, CE, dish_categories, coef, dtype)
    users = _as_ids(users, PMf.shape[0], "user")
    items = _as_ids(items, Dt.shape[0], "item")
    with np.errstate(invalid="ignore"):
        return np.einsum("bk,bk->b", PMf[users].reshape(len(users), -1), Dt[items])


# ----------------------------------------------------------------------------------------------
# Memory write (training side) -- Model.Write_Memory, Model_Recommender.py:106-220, op for op including the
# dense one-hot matmuls (small sizes only).  PARITY UNPINNED (TF arithmetic).  General_Memory is read
# before it is written: the two PM assigns and the GM assign all consume the pre-call value (:168, :215).
# ----------------------------------------------------------------------------------------------

def write_memory(PM, RE, CE, GM, users, items, categories, write_sign, user_one_hot_label,
                 beta_1=0.01, beta_2=0.01, alpha=0.01, dtype=np.float64, personal=True, general=True):
    """Returns (Personal_Memory', General_Memory', mean(PM'), mean(GM')).

    `personal` / `general` say which of the two fetches is in the sess.run list: TF executes only the assigns a fetch
    depends on -- `personal` = reduce_mean of the second Personal_Memory assign (:198, chained on :167), `general` =
    reduce_mean of the General_Memory assign (:215).  A table whose fetch is absent comes back unchanged."""
    PM = np.asarray(PM, dtype=dtype); RE = np.asarray(RE, dtype=dtype); CE = np.asarray(CE, dtype=dtype)
    GM = np.asarray(GM, dtype=dtype)
    U, C1, E = PM.shape
    C = C1 - 1
    users = _as_ids(users, U, "user"); items = _as_ids(items, RE.shape[0], "item")
    B = len(users)
    cat = _as_mask(categories, C).astype(dtype)                                   # [B, C, 1]
    s = np.asarray(write_sign, dtype=dtype).reshape(B, 1)                         # [B, 1]   :30
    y = np.asarray(user_one_hot_label, dtype=dtype).reshape(B, -1)                # [B, L]   :33
    with np.errstate(divide="ignore", invalid="ignore"):
        item_embedding = RE[items][:, None, :]                                    # :107-109
        dish_memory = cat * item_embedding                                        # :111
        dish_memory = dish_memory * (beta_1 * s)[:, None, :]                      # :115-119
        dish_category = (cat * CE).sum(axis=1)                                    # :124-128
        category_num = cat.sum(axis=(1, 2))[:, None]                              # :130-132
        dish_category = dish_category / category_num                              # :134
        dish_category = dish_category[:, None, :] * (beta_2 * s)[:, None, :]      # :138-145
        dish_memory = dish_memory.reshape(B, 1, C * E)                            # :149
        onehot = np.zeros((B, U, 1), dtype=dtype); onehot[np.arange(B), users, 0] = 1   # :151
        dish_bias = (onehot @ dish_memory).sum(axis=0).reshape(U, C, E)           # :154-156
        category_bias = (onehot @ dish_category).sum(axis=0)[:, None, :]          # :158-162
        PM1 = PM + np.concatenate([category_bias, dish_bias], axis=1)             # :164-167
        general_memory = GM.reshape(GM.shape[0], C1 * E)                          # :172
        ulm = (y[:, :, None] * general_memory).sum(axis=1)                        # :174-178
        ulm = ulm / y.sum(axis=1)[:, None]                                        # :180-184
        general_bias = (onehot @ ulm[:, None, :]).sum(axis=0).reshape(U, C1, E)   # :188-192
        PM2 = PM1 + alpha * general_bias                                          # :194-197
        ylab = y[:, :, None]                                                      # :169
        dgb = (ylab @ dish_memory).sum(axis=0).reshape(-1, C, E)                  # :200-205
        cgb = (ylab @ dish_category).sum(axis=0)[:, None, :]                      # :207-212
        GM2 = GM + np.concatenate([cgb, dgb], axis=1)                             # :213-215
    if not personal:
        PM2 = PM
    if not general:
        GM2 = GM
    return PM2, GM2, PM2.mean(), GM2.mean()


# ----------------------------------------------------------------------------------------------
# Build-defined extensions (NO reference counterpart; BASELINE.json configs 2-5).  These restate the
# build's own definitions; nothing in the reference pins them.
# ----------------------------------------------------------------------------------------------

def dish_high_vectors(ING, offsets, ids, weights=None, dtype=np.float64) -> np.ndarray:
    """H[d] = sum_j w_j ING[id_j] / sum_j w_j over the CSR list of dish d (empty list -> NaN row)."""
    ING = np.asarray(ING, dtype=dtype)
    offsets = np.asarray(offsets, dtype=np.int64)
    ids = np.asarray(ids, dtype=np.int64)
    w = np.ones(len(ids), dtype=dtype) if weights is None else np.asarray(weights, dtype=dtype)
    I = len(offsets) - 1
    H = np.empty((I, ING.shape[1]), dtype=dtype)
    with np.errstate(divide="ignore", invalid="ignore"):
        for d in range(I):
            sl = slice(offsets[d], offsets[d + 1])
            acc = np.zeros(ING.shape[1], dtype=dtype)
            tot = dtype(0)
            for j, wj in zip(ids[sl], w[sl]):                 # sequential, like the kernel
                acc = acc + wj * ING[j]
                tot = tot + wj
            H[d] = acc / tot
    return H


def inference_ingredients(PM, RE, ING, offsets, ids, weights, users, items, categories,
                          coef: float = DEFAULT_COEF, dtype=np.float64) -> np.ndarray:
    """score = a * <U_high, H[d]> + (1-a) * low, with `low` exactly as Model_Recommender.py:82-92."""
    PM = np.asarray(PM)
    C = PM.shape[1] - 1
    users = _as_ids(users, PM.shape[0], "user")
    items = _as_ids(items, np.asarray(RE).shape[0], "item")
    cat = _as_mask(categories, C).astype(dtype)
    PMd, REd = PM.astype(dtype), np.asarray(RE).astype(dtype)
    H = dish_high_vectors(ING, offsets, ids, weights, dtype)
    a32, b32 = blend_coefficients(coef)
    with np.errstate(divide="ignore", invalid="ignore"):
        UM = PMd[users]
        high = (UM[:, 0, :] * H[items]).sum(axis=1, dtype=dtype)
        n = cat.sum(axis=(1, 2), dtype=dtype)
        low = (REd[items][:, None, :] * (cat * UM[:, 1:, :])).sum(axis=(1, 2), dtype=dtype) / n
        return dtype(a32) * high + dtype(b32) * low


def inference_mlp(PM, RE, CE, dish_categories, W1, b1, W2, b2, w3, b3, users, items,
                  coef: float = DEFAULT_COEF, dtype=np.float64, dish_high=None) -> np.ndarray:
    """Build-defined 3-layer head: z = flatten(PM[u]) * Dt[d];
    score = sum(z) + w3 . relu(W2^T relu(W1^T z + b1) + b2) + b3.  `dish_high` [I, E] replaces the
    category high-level vector when the ingredient extension is active."""
    PMd = np.asarray(PM, dtype=dtype)
    Dt = dish_vectors(RE, CE, dish_categories, coef, dtype)
    if dish_high is not None:
        E = PMd.shape[2]
        Dt[:, :E] = dtype(blend_coefficients(coef)[0]) * np.asarray(dish_high, dtype=dtype)
    users = _as_ids(users, PMd.shape[0], "user")
    items = _as_ids(items, Dt.shape[0], "item")
    with np.errstate(invalid="ignore"):
        z = PMd[users].reshape(len(users), -1) * Dt[items]
        h1 = np.maximum(z @ np.asarray(W1, dtype=dtype) + np.asarray(b1, dtype=dtype), 0)
        h2 = np.maximum(h1 @ np.asarray(W2, dtype=dtype) + np.asarray(b2, dtype=dtype), 0)
        return z.sum(axis=1) + (h2 @ np.asarray(w3, dtype=dtype) + dtype(b3))


# ----------------------------------------------------------------------------------------------
# Evaluator (evaluate.py)
# ----------------------------------------------------------------------------------------------

def getHitRatio(ranklist: Sequence, gtItem) -> int:
    """evaluate.py:69-73."""
    for item in ranklist:
        if item == gtItem:
            return 1
    return 0


def getNDCG(ranklist: Sequence, gtItem) -> float:
    """evaluate.py:76-81 -- ln2 / ln(rank + 2), 0 when absent."""
    for i in range(len(ranklist)):
        if ranklist[i] == gtItem:
            return math.log(2) / math.log(i + 2)
    return 0


def candidate_batch(user, testRatings: Dict[str, List[int]], testNegatives: Dict[str, List[int]]):
    """evaluate.py:39-51 -- [positive] + negatives[50:100]; the user key is ``str(user)``."""
    positive = testRatings[str(user)][0]
    return [positive] + list(testNegatives[str(user)][50:100])


def rank_candidates(items: Sequence, scores: Sequence, K: int) -> List:
    """evaluate.py:53, :60-63 -- dict collapse (a repeated item keeps its FIRST position and its
    LAST score) followed by ``heapq.nlargest(K, dict, key=dict.get)``."""
    table = {}
    for it, sc in zip(items, scores):
        table[it] = sc
    return heapq.nlargest(K, table, key=table.get)


def eval_one_rating(score_fn, user, testRatings, testNegatives, K, dish_to_category):
    """evaluate.py:35-66 with ``score_fn(users, items, categories) -> scores`` in place of
    ``sess.run([model.logits], feed_dict)[0]`` (:55-59)."""
    if str(user) not in testRatings or len(testRatings[str(user)]) == 0:
        return None                                                    # :37-38
    items = candidate_batch(user, testRatings, testNegatives)
    users = [user] * len(items)
    cats = [dish_to_category[str(i)] for i in items]                   # :43, :50
    scores = score_fn(users, items, cats)
    ranklist = rank_candidates(items, scores, K)
    return getHitRatio(ranklist, items[0]), getNDCG(ranklist, items[0])


def evaluate_model(score_fn, testRatings, testNegatives, K, dish_to_category):
    """evaluate.py:13-32 -- iterate the users in dict order, one scoring call per user."""
    hits, ndcgs = [], []
    for idx in testRatings:
        hr, ndcg = eval_one_rating(score_fn, idx, testRatings, testNegatives, K, dish_to_category)
        hits.append(hr)
        ndcgs.append(ndcg)
    return hits, ndcgs


def topk_catalogue(PM, RE, CE, dish_categories, users: Iterable[int], k: int,
                   coef: float = DEFAULT_COEF, dtype=np.float64):
    """Full-catalogue retrieval (build-defined generalisation of evaluate.py:39-63 to every dish):
    score every dish with `inference`, rank with the reference's ``heapq.nlargest`` rule --
    descending score, ties to the earlier (lower) dish id.  NaN scores (0/0, :79/:92) rank last."""
    PM = np.asarray(PM)
    I = np.asarray(RE).shape[0]
    m = _as_mask(dish_categories, np.asarray(CE).shape[0])
    all_items = np.arange(I)
    out_s, out_i = [], []
    for u in users:
        s = inference(PM, RE, CE, np.full(I, int(u)), all_items, m, coef, dtype)
        key = np.where(np.isnan(s), -np.inf, s)
        order = np.lexsort((all_items, -key))[:k]
        out_i.append(order.astype(np.int64))
        out_s.append(s[order])
    return np.asarray(out_s), np.asarray(out_i)
