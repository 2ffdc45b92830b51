"""Host-side mirror of the reference's model-build / predict surface.

Reference: ``Code/Recommender/Model_Recommender.py`` (``Model.__init__`` :5-41, ``inference`` :56-97)
and its one call site ``sess.run([model.logits], feed_dict)`` (``evaluate.py:55-59``).  The same
constructor signature and attribute names are kept so reference-style driver code reads the same;
the graph itself is replaced by one fused HIP kernel behind ``include/m2d.h``.

The forward (scoring) path is the product.  The training-side fetches of the
reference's driver (Train_recommender.py:180-199) -- ``loss_value``, ``learning_rate``, ``train_op``
(Model_Recommender.py:99-104, :223-241; section 8f row N4) and ``personal`` / ``general``
(``Write_Memory``, :106-220; row N2) -- are served by the same engine through ``Session.run``.
"""
from __future__ import annotations

import itertools
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


class _ScalarFetch(float):
    """``model.learning_rate``: a number (args.lr) that can also be named in ``sess.run`` fetches
    (Model_Recommender.py:11, :227; Train_recommender.py:190)."""


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
        try:
            a = np.array(x, dtype=np.int64).reshape(-1)     # ints and decimal strings alike, one C loop
        except (ValueError, TypeError, OverflowError):
            a = np.fromiter((int(v) for v in x), dtype=np.int64, count=len(x))
    if a.size and (a.min() < -(2 ** 31) or a.max() >= 2 ** 31):
        raise IndexError("%s id does not fit int32" % what)
    return a.astype(np.int32)


def _mask(categories, C: int, B: int):
    """[B, C, 1] nested lists (dish_to_category.json values, evaluate.py:43) or [B, C] -> f32 [B, C]."""
    if isinstance(categories, torch.Tensor):
        m = categories.to(torch.float32)
    elif (isinstance(categories, list) and len(categories) == B and B > 0 and isinstance(categories[0], list)
          and len(categories[0]) == C and isinstance(categories[0][0], list)):
        # the reference's own feed, [B][C][1] Python lists: flattening through iterators is ~5x faster than
        # np.asarray on three levels of nesting (this conversion was half the cost of a 51-pair predict call)
        it = itertools.chain.from_iterable(itertools.chain.from_iterable(categories))
        try:
            m = np.fromiter(it, dtype=np.float32, count=B * C).reshape(B, C)
        except (ValueError, TypeError):
            m = None
        if m is None or next(it, None) is not None or sum(map(len, categories)) != B * C:
            m = np.asarray(categories, dtype=np.float32)    # ragged or deeper than expected: let numpy say what is wrong
    else:
        m = np.asarray(categories, dtype=np.float32)
    if m.ndim == 3 and m.shape[2] == 1:
        m = m.reshape(m.shape[0], m.shape[1])
    if m.ndim != 2 or m.shape[1] != C or m.shape[0] != B:
        raise ValueError("categories must be [B=%d, C=%d, 1] or [B, C]; got %r" % (B, C, tuple(m.shape)))
    return m


def _float_rows(x, B: int, width) -> np.ndarray:
    """[B][width] nested lists (the driver's label / one-hot / write-sign feeds) or arrays -> float32 ndarray.  Lists of
    lists go through one flat iterator (a third faster than np.asarray on two levels of nesting)."""
    if isinstance(x, list) and len(x) == B and B > 0 and isinstance(x[0], list):
        try:
            a = np.fromiter(itertools.chain.from_iterable(x), dtype=np.float32, count=-1)
            w_ = a.size // B
            if a.size == B * w_ and (width is None or w_ == width) and set(map(len, x)) == {w_}:
                return a.reshape(B, w_)
        except (ValueError, TypeError):
            pass
    return np.asarray(x, dtype=np.float32)


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
        # training-side fetches (Model_Recommender.py:38-41, :13); Global_Step never moves in the reference
        # (apply_gradients is called without it, :240), so learning_rate stays args.lr
        self.loss_value = _Fetch("loss_value")
        self.train_op = _Fetch("train_op")
        self.personal = _Fetch("personal")
        self.general = _Fetch("general")
        self.learning_rate = _ScalarFetch(0.001 if self.learning_rate is None else self.learning_rate)   # --lr default
        self.global_step = 0
        self.epoch_step = 0
        self.epoch_increment = _Fetch("epoch_increment")
        self._train_started = False

        pm_shape = tuple(Personal_Memory.shape)
        if len(pm_shape) != 3 or pm_shape[1] != self.num_categories + 1 or pm_shape[2] != self.embed_size:
            raise ValueError("Personal_Memory %r does not match num_categories=%d, embed_size=%d"
                             % (pm_shape, self.num_categories, self.embed_size))
        if user_base == 0 and pm_shape[0] != self.num_users:
            raise ValueError("Personal_Memory has %d users, args.num_users = %d" % (pm_shape[0], self.num_users))
        self.engine = ScoringEngine(Personal_Memory, Recipe_Embedding, Category_Embedding,
                                    coef=self.high_level_score_coefficient, device=device, user_base=user_base)
        self.device = self.engine.device
        self._gm_dev = None             # General_Memory on the device, created by the first personal / general fetch

    # -- predict ------------------------------------------------------------------------------------
    def predict_device(self, user_input, item_input, categories) -> torch.Tensor:
        """Scores as a device tensor, no synchronisation (stream-ordered)."""
        ut, dt, mt = self._feeds(user_input, item_input, categories)
        return torch.ops.m2d.score_pairs(self.engine.id, ut, dt, mt)

    def predict(self, user_input, item_input, categories, **ignored) -> np.ndarray:
        """``sess.run([model.logits], feed_dict)[0]`` (evaluate.py:55-59): float32 ``ndarray [B]``.

        Extra reference feeds (``labels``, ``dropout_keep_prob``, ``is_training_flag`` ...) are accepted
        and ignored: no op in the reference's forward reads them.  An out-of-range id raises
        ``IndexError`` (TF-CPU ``GatherV2`` raises ``InvalidArgumentError``)."""
        if any(isinstance(x, torch.Tensor) for x in (user_input, item_input, categories)):
            out = self.predict_device(user_input, item_input, categories)      # torch custom op, stream-ordered
            self.engine.check()
            return out.cpu().numpy()
        # host data (the reference's own feeds): one staged copy in, one out (m2d_score_pairs_host)
        u, d = _ids(user_input, "user"), _ids(item_input, "item")
        if len(d) != len(u):
            raise ValueError("user_input and item_input differ in length")
        return self.engine.score_pairs_host(u, d, _mask(categories, self.num_categories, len(u)))

    # -- training side -----------------------------------------------------
    def _feeds(self, user_input, item_input, categories):
        u, d = _ids(user_input, "user"), _ids(item_input, "item")
        B = len(u)
        if len(d) != B:
            raise ValueError("user_input and item_input differ in length")
        m = _mask(categories, self.num_categories, B)
        dev = self.device
        ut = u.to(dev, torch.int32) if isinstance(u, torch.Tensor) else torch.from_numpy(u).to(dev)
        dt = d.to(dev, torch.int32) if isinstance(d, torch.Tensor) else torch.from_numpy(d).to(dev)
        mt = m.to(dev) if isinstance(m, torch.Tensor) else torch.from_numpy(m).to(dev)
        return ut, dt, mt

    def _device_batch(self, user_input, item_input, categories, floats):
        """Host feeds of one driver batch (Train_recommender.py:189-194) -> device tensors through ONE staged copy:
        users | items | masks | each entry of `floats` (name -> (feed, width)) packed into a single buffer.  Returns
        (users, items, masks, {name: tensor [B, width]}); feeds that are already tensors take `_feeds`."""
        extras = [(k, v, wd) for k, (v, wd) in floats.items()]
        if any(isinstance(x, torch.Tensor) for x in (user_input, item_input, categories)) or \
                any(isinstance(v, torch.Tensor) for _, v, _ in extras):
            ut, dt, mt = self._feeds(user_input, item_input, categories)
            B = ut.numel()
            out = {}
            for k, v, wd in extras:
                t = v if isinstance(v, torch.Tensor) else torch.as_tensor(_float_rows(v, B, wd))
                out[k] = t.to(self.device, torch.float32).reshape(B, t.numel() // B if B else 0)
            return ut, dt, mt, out
        u, d = _ids(user_input, "user"), _ids(item_input, "item")
        B = len(u)
        if len(d) != B:
            raise ValueError("user_input and item_input differ in length")
        m = _mask(categories, self.num_categories, B)
        C = self.num_categories
        rows = [_float_rows(v, B, wd) for _, v, wd in extras]
        up = lambda n: (n + 3) & ~3                            # every section starts on a 16-byte boundary
        Bp = up(B)
        o_m = 2 * Bp
        pos = o_m + up(B * C)
        spans = []
        for r in rows:
            spans.append((pos, r.size))
            pos += up(r.size)
        buf = np.zeros(pos, dtype=np.float32)
        ib = buf.view(np.int32)
        ib[:B] = u
        ib[Bp:Bp + B] = d
        buf[o_m:o_m + B * C] = m.reshape(-1)
        for r, (a, n) in zip(rows, spans):
            buf[a:a + n] = r.reshape(-1)
        t = torch.from_numpy(buf).to(self.device)
        it = t.view(torch.int32)
        out = {k: t[a:a + n].reshape(B, n // B if B else 0) for (k, _, _), (a, n) in zip(extras, spans)}
        return it[:B], it[Bp:Bp + B], t[o_m:o_m + B * C].reshape(B, C), out

    def train_step(self, user_input, item_input, categories, labels, apply: bool = True):
        """``sess.run([model.loss_value, model.learning_rate, model.train_op], feed_dict)``
        (Train_recommender.py:189-199): returns ``(loss, learning_rate)`` as Python floats; with ``apply=False``
        only the loss is evaluated.  The optimizer is ``args.learner`` at ``args.lr`` (Model_Recommender.py:223-241)."""
        if not self._train_started:
            self.engine.train_begin(self.learner or "sgd", float(self.learning_rate), 5.0)
            self._train_started = True
        ut, dt, mt = self._feeds(user_input, item_input, categories)
        y = labels if isinstance(labels, torch.Tensor) else torch.as_tensor(np.asarray(labels, dtype=np.float32))
        out = self.engine.train_step(ut, dt, mt, y.reshape(-1), apply=apply)
        self.engine.check()
        loss, _norm, _scale, lr = (float(v) for v in out.cpu().numpy())
        return loss, lr

    def write_memory(self, user_input, item_input, categories, write_sign, user_one_hot_label,
                     personal: bool = True, general: bool = True):
        """The ``personal`` / ``general`` fetches (``Write_Memory``, Model_Recommender.py:106-220).  As in the reference
        graph each fetch pulls in only the assigns it depends on: ``personal`` the two chained Personal_Memory assigns
        (:167, :198), ``general`` the General_Memory assign (:215).  Updates those tables in place and returns
        ``(mean(Personal_Memory), mean(General_Memory))`` with ``None`` for a table that was not fetched."""
        if self._gm_dev is None:
            if self.General_Memory is None:
                raise ValueError("Model was built without General_Memory")
            self._gm_dev = torch.as_tensor(np.asarray(self.General_Memory, dtype=np.float32)).to(self.device).contiguous()
        ut, dt, mt = self._feeds(user_input, item_input, categories)
        B = ut.numel()
        as_t = lambda v, wd: v if isinstance(v, torch.Tensor) else torch.as_tensor(_float_rows(v, B, wd))
        sign = as_t(write_sign, 1).reshape(B)
        y = as_t(user_one_hot_label, None).reshape(B, -1)
        return self.engine.write_memory(ut, dt, mt, sign, y, self._gm_dev, float(self.beta_1), float(self.beta_2),
                                        float(self.alpha), want_means=True, write_pm=personal, write_gm=general)

    # -- checkpoint (stands where the driver uses tf.train.Saver, Train_recommender.py:145-149, :218-222) ----------
    _TABLE_FILES = ("Personal_Memory", "Recipe_Embedding", "Category_Embedding", "General_Memory")

    def save(self, path: str):
        """One ``.npz``: the four tables under the names of the reference's ``.npy`` files (so each can be re-saved
        with ``np.save`` and fed back to the reference), plus the optimizer's slots and step count once training has
        started."""
        self.engine.check()
        data = {"Personal_Memory": self.engine.pm.cpu().numpy(), "Recipe_Embedding": self.engine.re.cpu().numpy(),
                "Category_Embedding": self.engine.ce.cpu().numpy(), "epoch_step": np.int64(self.epoch_step)}
        if self.General_Memory is not None:
            data["General_Memory"] = self.general_memory()
        if self._train_started:
            data["learner"] = np.str_(str(self.learner or "sgd").lower())
            data["steps"] = np.int64(self.engine.train_steps())
            for tb in range(3):
                for sl in range(2):
                    try:
                        data["slot_%d_%d" % (tb, sl)] = self.engine.train_slot(tb, sl).cpu().numpy()
                    except ValueError:                      # this learner has no such slot
                        pass
        np.savez(path, **data)

    def restore(self, path: str):
        """Load what `save` wrote INTO this model (same shapes): tables in place on the device, optimizer state if the
        file has it and the learner matches."""
        z = np.load(path if str(path).endswith(".npz") else str(path) + ".npz")
        for name, dst in (("Personal_Memory", self.engine.pm), ("Recipe_Embedding", self.engine.re),
                          ("Category_Embedding", self.engine.ce)):
            src = torch.from_numpy(z[name])
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError("%s in the checkpoint is %r, the model has %r" % (name, tuple(src.shape), tuple(dst.shape)))
            dst.copy_(src)
        self.engine.tables_updated()
        if "General_Memory" in z.files:
            self.General_Memory = z["General_Memory"]
            self._gm_dev = None
        self.epoch_step = int(z["epoch_step"]) if "epoch_step" in z.files else 0
        if "steps" in z.files:
            if str(z["learner"]) != str(self.learner or "sgd").lower():
                raise ValueError("checkpoint was trained with %s, this model uses %s" % (z["learner"], self.learner))
            self.engine.train_begin(self.learner or "sgd", float(self.learning_rate), 5.0)
            self._train_started = True
            self.engine.train_steps(restore=int(z["steps"]))
            for key in z.files:
                if key.startswith("slot_"):
                    _, tb, sl = key.split("_")
                    self.engine.train_slot(int(tb), int(sl), restore=torch.from_numpy(z[key]))
        self.engine.check()

    def general_memory(self) -> np.ndarray:
        """General_Memory as it stands (host copy)."""
        return np.asarray(self.General_Memory, dtype=np.float32) if self._gm_dev is None else self._gm_dev.cpu().numpy()

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

    def run(self, fetches, feed_dict=None):
        """Fetches served: ``logits`` (evaluate.py:58); ``loss_value``, ``learning_rate``, ``train_op``, ``personal``,
        ``general`` (Train_recommender.py:180-199); ``epoch_increment`` / ``epoch_step`` (:155, :204).

        The reference fetches the memory write and ``train_op`` in one ``sess.run`` with no control dependency
        between them, so TF may order them either way; here the optimizer step runs first and ``Write_Memory``
        sees the updated tables.  ``general`` alone (the driver's ordinary batch, Train_recommender.py:195-199) writes
        General_Memory only; Personal_Memory is written only when ``personal`` is fetched (:180-184)."""
        as_list = isinstance(fetches, (list, tuple))
        fl: Sequence = fetches if as_list else [fetches]
        m = self.model
        known = (m.logits, m.loss_value, m.train_op, m.personal, m.general, m.epoch_increment, m.learning_rate)
        for f in fl:
            if not any(f is k for k in known) and f != "epoch_step":
                raise NotImplementedError("fetch %r is not served" % (f,))
        feed_dict = feed_dict or {}

        def feed(key, what):
            try:
                return feed_dict[key]
            except KeyError:
                raise ValueError("feed_dict is missing %r" % (what,)) from None

        has = lambda k: any(f is k for f in fl)
        res = {}
        if has(m.logits):
            res[id(m.logits)] = m.predict(feed(m.user_input, m.user_input), feed(m.item_input, m.item_input),
                                          feed(m.categories, m.categories))
        need_train = has(m.train_op) or has(m.loss_value)
        need_write = has(m.personal) or has(m.general)
        if need_train or need_write:
            # the batch crosses to the device once, whatever is fetched (Train_recommender.py:189-199 feeds six lists)
            floats = {}
            if need_train:
                floats["labels"] = (feed(m.labels, m.labels), 1)
            if need_write:
                floats["sign"] = (feed(m.write_sign, m.write_sign), 1)
                floats["onehot"] = (feed(m.user_one_hot_label, m.user_one_hot_label), None)
            ut, dt, mt, ex = m._device_batch(feed(m.user_input, m.user_input), feed(m.item_input, m.item_input),
                                             feed(m.categories, m.categories), floats)
        if need_train:
            loss, lr = m.train_step(ut, dt, mt, ex["labels"], apply=has(m.train_op))
            res[id(m.loss_value)] = np.float32(loss)
            res[id(m.train_op)] = None
            res["lr"] = np.float32(lr)
        if need_write:
            pmean, gmean = m.write_memory(ut, dt, mt, ex["sign"], ex["onehot"],
                                          personal=has(m.personal), general=has(m.general))
            res[id(m.personal)] = None if pmean is None else np.float32(pmean)
            res[id(m.general)] = None if gmean is None else np.float32(gmean)
        out = []
        for f in fl:
            if f is m.learning_rate:
                out.append(res.get("lr", np.float32(m.learning_rate)))
            elif f is m.epoch_increment:
                m.epoch_step += 1
                out.append(m.epoch_step)
            elif isinstance(f, str) and f == "epoch_step":
                out.append(m.epoch_step)
            else:
                out.append(res[id(f)])
        return out if as_list else out[0]
