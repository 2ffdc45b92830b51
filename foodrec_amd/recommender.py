"""Host-side mirror of the reference's model-build / predict surface.

Reference: ``Code/Recommender/Model_Recommender.py`` (``Model.__init__`` :5-41, ``inference`` :56-97)
and its one call site ``sess.run([model.logits], feed_dict)`` (``evaluate.py:55-59``).  The same
constructor signature and attribute names are kept so reference-style driver code reads the same;
the graph itself is replaced by one fused HIP kernel behind ``include/m2d.h``.

Only the forward (scoring) path exists here.  ``loss`` / ``Write_Memory`` / ``train``
(Model_Recommender.py:99-241) are out of scope.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from .ops import ScoringEngine


class _Placeholder:
    """Stands where the reference has a ``tf.placeholder`` (Model_Recommender.py:26-35): a feed_dict key."""

    def __init__(self, name: str, used: bool):
        self.name = name
        self.affects_logits = used

    def __repr__(self):
        return "<placeholder %s>" % self.name


class _Fetch:
    def __init__(self, name: str):
        self.name = name

    def __repr__(self):
        return "<fetch %s>" % self.name


def _ids(x, what: str) -> np.ndarray:
    """int32 feed conversion; ids reach the reference as Python ``str`` keys (evaluate.py:28, :41)."""
    if isinstance(x, torch.Tensor):
        return x
    if isinstance(x, np.ndarray) and x.dtype.kind in "iu":
        a = x.astype(np.int64, copy=False).reshape(-1)
    else:
        a = np.fromiter((int(v) for v in x), dtype=np.int64, count=len(x))
    if a.size and (a.min() < -(2 ** 31) or a.max() >= 2 ** 31):
        raise IndexError("%s id does not fit int32" % what)
    return a.astype(np.int32)


def _mask(categories, C: int, B: int):
    """[B, C, 1] nested lists (dish_to_category.json values, evaluate.py:43) or [B, C] -> f32 [B, C]."""
    if isinstance(categories, torch.Tensor):
        m = categories.to(torch.float32)
    else:
        m = np.asarray(categories, dtype=np.float32)
    if m.ndim == 3 and m.shape[2] == 1:
        m = m.reshape(m.shape[0], m.shape[1])
    if m.ndim != 2 or m.shape[1] != C or m.shape[0] != B:
        raise ValueError("categories must be [B=%d, C=%d, 1] or [B, C]; got %r" % (B, C, tuple(m.shape)))
    return m


class Model:
    """``Model(args, Personal_Memory, Recipe_Embedding, Category_Embedding, General_Memory)``.

    ``args`` needs the attributes the reference constructor reads (Model_Recommender.py:6-24):
    ``num_categories, num_users, embed_size, high_level_score_coefficient`` are used; ``learner,
    num_labels, lr, decay_steps, decay_rate, beta_1, beta_2, alpha`` are recorded when present.
    Tables are numpy (or torch) arrays and are copied to HBM once; the caller keeps its host copies.
    ``General_Memory`` is accepted for signature parity; the forward never reads it.
    """

    def __init__(self, args, Personal_Memory, Recipe_Embedding, Category_Embedding, General_Memory=None,
                 device: Optional[torch.device] = None, user_base: int = 0):
        for name in ("learner", "num_labels", "lr", "decay_steps", "decay_rate", "beta_1", "beta_2", "alpha"):
            setattr(self, name if name != "lr" else "learning_rate", getattr(args, name, None))
        self.num_categories = int(args.num_categories)
        self.num_users = int(args.num_users)
        self.embed_size = int(args.embed_size)
        self.high_level_score_coefficient = float(np.float32(args.high_level_score_coefficient))
        self.General_Memory = General_Memory

        # the input contract (Model_Recommender.py:26-35); only three feeds reach the logits
        self.user_input = _Placeholder("user_input", True)
        self.item_input = _Placeholder("item_input", True)
        self.categories = _Placeholder("categories", True)
        self.labels = _Placeholder("labels", False)
        self.write_sign = _Placeholder("write_sign", False)
        self.user_one_hot_label = _Placeholder("user_labels", False)
        self.dropout_keep_prob = _Placeholder("dropout_keep_prob", False)
        self.is_training_flag = _Placeholder("is_training_flag", False)
        self.logits = _Fetch("logits")

        pm_shape = tuple(Personal_Memory.shape)
        if len(pm_shape) != 3 or pm_shape[1] != self.num_categories + 1 or pm_shape[2] != self.embed_size:
            raise ValueError("Personal_Memory %r does not match num_categories=%d, embed_size=%d"
                             % (pm_shape, self.num_categories, self.embed_size))
        if user_base == 0 and pm_shape[0] != self.num_users:
            raise ValueError("Personal_Memory has %d users, args.num_users = %d" % (pm_shape[0], self.num_users))
        self.engine = ScoringEngine(Personal_Memory, Recipe_Embedding, Category_Embedding,
                                    coef=self.high_level_score_coefficient, device=device, user_base=user_base)
        self.device = self.engine.device

    # -- predict ------------------------------------------------------------------------------------
    def predict_device(self, user_input, item_input, categories) -> torch.Tensor:
        """Scores as a device tensor, no synchronisation (stream-ordered)."""
        u, d = _ids(user_input, "user"), _ids(item_input, "item")
        B = len(u)
        if len(d) != B:
            raise ValueError("user_input and item_input differ in length")
        m = _mask(categories, self.num_categories, B)
        dev = self.device
        ut = u.to(dev, torch.int32) if isinstance(u, torch.Tensor) else torch.from_numpy(u).to(dev)
        dt = d.to(dev, torch.int32) if isinstance(d, torch.Tensor) else torch.from_numpy(d).to(dev)
        mt = m.to(dev) if isinstance(m, torch.Tensor) else torch.from_numpy(m).to(dev)
        return torch.ops.m2d.score_pairs(self.engine.id, ut, dt, mt)

    def predict(self, user_input, item_input, categories, **ignored) -> np.ndarray:
        """``sess.run([model.logits], feed_dict)[0]`` (evaluate.py:55-59): float32 ``ndarray [B]``.

        Extra reference feeds (``labels``, ``dropout_keep_prob``, ``is_training_flag`` ...) are accepted
        and ignored: no op in the reference's forward reads them.  An out-of-range id raises
        ``IndexError`` (TF-CPU ``GatherV2`` raises ``InvalidArgumentError``)."""
        out = self.predict_device(user_input, item_input, categories)
        self.engine.check()
        return out.cpu().numpy()

    # -- resident dish -> category table (dish_to_category.json) --------------------------------------
    def set_dish_categories(self, dish_to_category, num_dishes: Optional[int] = None):
        """Accepts the JSON dict ``{str(dish): [[m0], [m1], ...]}`` (Train_recommender.py:132) or an
        ``[I, C]`` array.  Dishes missing from the dict get an all-zero mask (their score is NaN, as a
        zero mask gives in the reference)."""
        I = self.engine.I if num_dishes is None else num_dishes
        if isinstance(dish_to_category, dict):
            table = np.zeros((I, self.num_categories), dtype=np.float32)
            for key, val in dish_to_category.items():
                d = int(key)
                if 0 <= d < I:
                    table[d] = np.asarray(val, dtype=np.float32).reshape(-1)
        else:
            table = np.asarray(dish_to_category, dtype=np.float32).reshape(I, self.num_categories)
        self.engine.set_dish_categories(table)
        return table


    # -- retrieval and build-defined extensions (no reference counterpart; DESIGN.md sections 4.3-4.4, 8) ---
    def topk(self, users, k: int = 10):
        """The k best dishes of every user in `users` over the whole catalogue -- the ranking rule of
        evaluate.py:63 applied to all dishes instead of 51 candidates.  Needs `set_dish_categories`.
        Returns (scores float32 [n, k], dish ids int32 [n, k])."""
        u = _ids(users, "user")
        ut = u.to(self.device, torch.int32) if isinstance(u, torch.Tensor) else torch.from_numpy(u).to(self.device)
        s, i = torch.ops.m2d.topk_users(self.engine.id, ut, int(k))
        self.engine.check()
        return s.cpu().numpy(), i.cpu().numpy()

    def set_ingredients(self, ingredient_table, offsets, ids, weights=None):
        """EXTENSION: multi-hot ingredient lists per dish (CSR) replacing the category sum of the high-level path."""
        self.engine.set_ingredients(ingredient_table, offsets, ids, weights)

    def set_mlp_head(self, W1, b1, W2, b2, w3, b3: float):
        """EXTENSION: 3-layer head added to the reference score (interaction vector -> 256 -> 64 -> 1)."""
        self.engine.set_mlp_head(W1, b1, W2, b2, w3, b3)

    def predict_extended(self, user_input, item_input, categories=None, head: bool = False) -> np.ndarray:
        """EXTENSION predict: ingredient high-level path (`head=False`; `categories=None` -> resident dish masks)
        or reference score + MLP head (`head=True`, resident dish masks)."""
        u, d = _ids(user_input, "user"), _ids(item_input, "item")
        dev = self.device
        ut = u.to(dev, torch.int32) if isinstance(u, torch.Tensor) else torch.from_numpy(u).to(dev)
        dt = d.to(dev, torch.int32) if isinstance(d, torch.Tensor) else torch.from_numpy(d).to(dev)
        if head:
            out = self.engine.score_pairs_mlp(ut, dt)
        else:
            m = None
            if categories is not None:
                m = _mask(categories, self.num_categories, len(u))
                m = m.to(dev) if isinstance(m, torch.Tensor) else torch.from_numpy(m).to(dev)
            out = self.engine.score_pairs_ingredients(ut, dt, m)
        self.engine.check()
        return out.cpu().numpy()


class Session:
    """Shim for ``tf.Session`` at the one place the scoring path uses it: ``sess.run(fetches, feed_dict)``
    with ``fetches`` = ``model.logits`` or ``[model.logits]`` (evaluate.py:58)."""

    def __init__(self, model: Model):
        self.model = model

    def run(self, fetches, feed_dict):
        as_list = isinstance(fetches, (list, tuple))
        fl: Sequence = fetches if as_list else [fetches]
        m = self.model
        for f in fl:
            if f is not m.logits:
                raise NotImplementedError("only model.logits can be fetched (forward path); got %r" % (f,))
        try:
            u, d, c = feed_dict[m.user_input], feed_dict[m.item_input], feed_dict[m.categories]
        except KeyError as e:
            raise ValueError("feed_dict is missing %r" % (e.args[0],)) from None
        scores = m.predict(u, d, c)
        out = [scores for _ in fl]
        return out if as_list else out[0]
