"""CPU oracle for the Market2Dish training step.  TEST INFRASTRUCTURE ONLY.

Only ``tests/`` may import this; ``foodrec_amd`` never does.

What it restates
    ``Model.loss``  (``Code/Recommender/Model_Recommender.py:99-104``) -- mean sigmoid cross-entropy of the
    logits of ``Model.inference`` (:56-97);
    ``Model.train`` (:223-241) -- ``optimizer.compute_gradients`` -> ``tf.clip_by_global_norm(gradients, 5.0)``
    -> ``optimizer.apply_gradients`` for ``learner`` in {adam, adagrad, rmsprop, anything else = SGD};
    the call site ``sess.run([model.loss_value, model.learning_rate, ..., model.train_op], feed_dict)``
    (``Train_recommender.py:180-199``).

PARITY UNPINNED.  The gradients, the clip and the four update rules execute inside TensorFlow 1.x, which is not
in this image; the reference ships no test or fixture for a training step.  The rules
below restate the PUBLISHED behaviour of the TF 1.x ops the reference calls, by name:

* ``tf.nn.sigmoid_cross_entropy_with_logits``: ``max(x, 0) - x z + log(1 + exp(-|x|))``; gradient
  ``sigmoid(x) - z``.
* The gradient of ``tf.nn.embedding_lookup`` on a variable is an ``IndexedSlices`` of per-pair rows (duplicate
  ids NOT yet summed).  ``tf.clip_by_global_norm`` takes the norm over those per-pair rows (``.values``) and the
  dense ``Category_Embedding`` gradient; ``General_Memory`` gets no gradient (the loss never reads it) and is
  skipped.  ``scale = clip_norm * min(1 / norm, 1 / clip_norm)``.
* ``Optimizer.apply_gradients`` sums duplicate slices first (``_apply_sparse_duplicate_indices``) and then:
    - Adam (``beta1 = 0.9, beta2 = 0.999, eps = 1e-8``): ``lr_t = lr sqrt(1 - b2^t) / (1 - b1^t)``; the sparse
      path decays ``m`` and ``v`` of EVERY row, scatter-adds ``(1 - b1) g`` / ``(1 - b2) g^2`` at the touched
      rows and then updates EVERY row with ``lr_t m / (sqrt(v) + eps)`` (``_apply_sparse_shared``);
    - Adagrad (accumulator starts at 0.1): touched rows only, ``acc += g^2; var -= lr g / sqrt(acc)``;
    - RMSProp (``decay 0.9, momentum 0, eps 1e-10``, ``rms`` slot starts at ONE): touched rows only,
      ``ms += (g^2 - ms)(1 - decay); mom = momentum mom + lr g / sqrt(ms + eps); var -= mom``;
    - SGD: ``var -= lr g``.
* ``apply_gradients`` is called WITHOUT ``global_step`` (:240), so ``Global_Step`` stays 0 and
  ``tf.train.exponential_decay(..., staircase=True)`` always returns ``lr * decay_rate ** 0 = lr``.

Everything is float32 in TF; this oracle computes in float64 from the float32 inputs (the tests bound the
difference) unless ``dtype=np.float32`` is asked for.
"""
from __future__ import annotations

import numpy as np

SGD, ADAGRAD, RMSPROP, ADAM = 0, 1, 2, 3
LEARNERS = {"sgd": SGD, "adagrad": ADAGRAD, "rmsprop": RMSPROP, "adam": ADAM}


def learner_code(name: str) -> int:
    """Model_Recommender.py:228-235: anything that is not adagrad / rmsprop / adam is plain gradient descent."""
    return LEARNERS.get(str(name).lower(), SGD)


def loss_and_gradients(PM, RE, CE, users, items, categories, labels, coef=0.99, dtype=np.float64):
    """Forward (Model_Recommender.py:56-97), loss (:99-104) and the gradients ``compute_gradients`` returns:
    per-pair rows for Personal_Memory [B, C+1, E] and Recipe_Embedding [B, E], dense for Category_Embedding."""
    a = np.float32(coef)
    b = np.float32(1.0) - a                                             # :96, float32
    a, b = dtype(a), dtype(b)
    users = np.asarray(users, dtype=np.int64)
    items = np.asarray(items, dtype=np.int64)
    m = np.asarray(categories, dtype=dtype).reshape(len(users), -1)     # [B, C]
    y = np.asarray(labels, dtype=dtype).reshape(-1)
    UM = np.asarray(PM, dtype=dtype)[users]                             # :57
    Uh, Ul = UM[:, 0, :], UM[:, 1:, :]                                  # :59
    It = np.asarray(RE, dtype=dtype)[items]                             # :63
    CEd = np.asarray(CE, dtype=dtype)
    n = m.sum(1)                                                        # :77
    H = m @ CEd                                                         # sum_c m_c CE_c     [B, E]
    L = np.einsum("bc,bce->be", m, Ul)                                  # sum_c m_c U_low,c  [B, E]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        s = a * ((Uh * H).sum(1) / n) + b * ((It * L).sum(1) / n)       # :79, :93, :95-96
        losses = np.maximum(s, 0) - s * y + np.log1p(np.exp(-np.abs(s)))   # :101
        loss = losses.mean()                                            # :103
        gs = (1.0 / (1.0 + np.exp(-s)) - y) / len(users)                # d loss / d s_b
        qh, ql = gs * a / n, gs * b / n
    dUM = np.empty_like(UM)
    dUM[:, 0, :] = qh[:, None] * H                                      # through :71-75
    dUM[:, 1:, :] = (ql[:, None] * m)[:, :, None] * It[:, None, :]      # through :82-90
    dIt = ql[:, None] * L
    dCE = (qh[:, None] * m).T @ Uh                                      # [C, E], summed over the batch
    return s, loss, dUM, dIt, dCE


class TrainState:
    """Tables + optimizer slots, stepping as ``sess.run([..., model.train_op])`` does.  ``GM`` is never touched
    by ``train_op``: its gradient is None."""

    def __init__(self, PM, RE, CE, learner="adam", lr=0.001, coef=0.99, clip_norm=5.0, dtype=np.float64):
        self.dtype = dtype
        self.PM, self.RE, self.CE = (np.array(t, dtype=dtype) for t in (PM, RE, CE))
        self.rule = learner_code(learner)
        self.lr = dtype(np.float32(lr))
        self.coef = coef
        self.clip = dtype(clip_norm)
        self.steps = 0
        z = lambda t: np.zeros_like(t)
        if self.rule == ADAM:
            self.slots = [[z(self.PM), z(self.PM)], [z(self.RE), z(self.RE)], [z(self.CE), z(self.CE)]]
            self.b1p, self.b2p = dtype(np.float32(0.9)), dtype(np.float32(0.999))     # beta powers, start at beta
        elif self.rule == ADAGRAD:
            self.slots = [[np.full_like(t, 0.1)] for t in (self.PM, self.RE, self.CE)]
        elif self.rule == RMSPROP:
            self.slots = [[np.ones_like(t), z(t)] for t in (self.PM, self.RE, self.CE)]
        else:
            self.slots = [[], [], []]

    def learning_rate(self):
        return self.lr                                                  # Global_Step never moves (:240)

    def step(self, users, items, categories, labels, apply=True):
        """Returns (loss, global_norm).  With apply=False nothing is updated (the loss_value fetch alone)."""
        dt = self.dtype
        s, loss, dUM, dIt, dCE = loss_and_gradients(self.PM, self.RE, self.CE, users, items, categories, labels,
                                                    self.coef, dt)
        norm = np.sqrt((dUM ** 2).sum() + (dIt ** 2).sum() + (dCE ** 2).sum())      # per-pair rows, not summed
        if not apply:
            return loss, norm
        with np.errstate(divide="ignore", invalid="ignore"):
            scale = self.clip * np.minimum(1.0 / norm, 1.0 / self.clip)
        users = np.asarray(users, dtype=np.int64)
        items = np.asarray(items, dtype=np.int64)
        grads = []
        for ids, vals, table in ((users, dUM, self.PM), (items, dIt, self.RE)):
            uniq, inv = np.unique(ids, return_inverse=True)
            g = np.zeros((len(uniq),) + vals.shape[1:], dtype=dt)
            np.add.at(g, inv, vals * scale)                              # duplicate slices summed
            grads.append((uniq, g))
        grads.append((None, dCE * scale))
        tables = (self.PM, self.RE, self.CE)
        lr = self.lr
        if self.rule == ADAM:
            b1, b2, eps = dt(np.float32(0.9)), dt(np.float32(0.999)), dt(np.float32(1e-8))
            lr_t = lr * np.sqrt(1 - self.b2p) / (1 - self.b1p)
            for (rows, g), var, (m, v) in zip(grads, tables, self.slots):
                m *= b1
                v *= b2
                if rows is None:
                    m += (1 - b1) * g
                    v += (1 - b2) * g * g
                else:
                    m[rows] += (1 - b1) * g
                    v[rows] += (1 - b2) * g * g
                var -= lr_t * m / (np.sqrt(v) + eps)                     # every row, touched or not
            self.b1p *= b1
            self.b2p *= b2
        else:
            for (rows, g), var, slots in zip(grads, tables, self.slots):
                idx = slice(None) if rows is None else rows
                if self.rule == ADAGRAD:
                    acc = slots[0]
                    acc[idx] += g * g
                    var[idx] -= lr * g / np.sqrt(acc[idx])
                elif self.rule == RMSPROP:
                    ms, mom = slots
                    ms[idx] += (g * g - ms[idx]) * (1 - dt(np.float32(0.9)))
                    mom[idx] = mom[idx] * 0.0 + lr * g / np.sqrt(ms[idx] + dt(np.float32(1e-10)))
                    var[idx] -= mom[idx]
                else:
                    var[idx] -= lr * g
        self.steps += 1
        return loss, norm
