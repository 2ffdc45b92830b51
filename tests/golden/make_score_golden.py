"""Generate the scoring-arithmetic golden vectors (tests/golden/score_*.npz).

These are produced by the build's OWN restatement (oracle/m2d_oracle.py), not by the reference:
TensorFlow is unavailable, so for rows A3-A7 the fixtures freeze the oracle (guarding it against
drift) rather than pin it to TF -- "parity unpinned", see oracle/m2d_oracle.py.  The one externally
checkable vector is the hand-computed KAT of SURVEY.md section 8a (score = 3.4625).

Each file: PM, RE, CE, users, items, cats (inputs, float32 / int32), score_f64 (float64 expression
tree), score_f32 (float32 op-for-op, materialised temporaries), coef.

The `*_coef*` cases (round 5) freeze the restatement at other values of `--high_level_score_coefficient`
(Train_recommender.py:61-62) than the flag's default 0.99: 0 (low level only), 0.5, 1 (high level only: `1 - coef`
is an exact float32 zero, Model_Recommender.py:96) and 1.25 (a negative low-level weight).

Usage:  python tests/golden/make_score_golden.py [--force]      (existing files are kept unless --force: an .npz
        carries zip time stamps, so a rewrite would change bytes without changing a value)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import m2d_oracle as oracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# (name, U, I, C, E, B, seed)
CASES = [
    ("E6", 11, 13, 4, 6, 37, 101),
    ("E32", 257, 129, 4, 32, 300, 102),
    ("E64", 96, 64, 4, 64, 333, 103),
    ("E128", 64, 48, 4, 128, 257, 104),
    ("E200", 40, 30, 4, 200, 130, 105),
    ("C3E8", 17, 9, 3, 8, 70, 106),
    ("C6E20", 12, 10, 6, 20, 65, 107),
    # name, U, I, C, E, B, seed, coef
    ("E32_coef000", 120, 90, 4, 32, 300, 108, 0.0),
    ("E64_coef050", 96, 64, 4, 64, 333, 109, 0.5),
    ("E200_coef100", 40, 30, 4, 200, 130, 110, 1.0),
    ("E128_coef125", 64, 48, 4, 128, 257, 111, 1.25),
]


def make_case(U, I, C, E, B, seed):
    rng = np.random.default_rng(seed)
    s = 1.0 / np.sqrt(E)
    PM = (rng.standard_normal((U, C + 1, E)) * s).astype(np.float32)
    RE = (rng.standard_normal((I, E)) * s).astype(np.float32)
    CE = (rng.standard_normal((C, E)) * s).astype(np.float32)
    users = rng.integers(0, U, B).astype(np.int32)
    items = rng.integers(0, I, B).astype(np.int32)
    cats = rng.integers(0, 2, (B, C)).astype(np.float32)
    cats[cats.sum(1) == 0, 0] = 1.0                       # random rows non-empty; edge rows set below
    # edge rows
    cats[0] = 0.0; cats[0, C - 1] = 1.0                   # single category
    cats[1] = 1.0                                         # every category
    cats[2] = 0.0                                         # no category -> 0/0 = NaN (:79, :92)
    cats[3] = rng.uniform(0.1, 2.5, C).astype(np.float32)  # non-binary weights (placeholder is float)
    users[4] = U - 1; items[4] = I - 1                    # last rows of both tables
    users[5] = 0; items[5] = 0
    return PM, RE, CE, users, items, cats


def main():
    force = "--force" in sys.argv
    keep = lambda name: os.path.exists(os.path.join(OUT, "score_%s.npz" % name)) and not force
    # hand KAT
    PM = np.array([[[1, 2], [1, 0], [0, 1], [2, 2], [3, -1]]], dtype=np.float32)
    RE = np.array([[0.5, -1]], dtype=np.float32)
    CE = np.array([[1, 1], [2, 0], [0, 2], [-1, 1]], dtype=np.float32)
    users = np.zeros(1, np.int32); items = np.zeros(1, np.int32)
    cats = np.array([[1, 0, 1, 0]], dtype=np.float32)
    if not keep("KAT"):
        np.savez(os.path.join(OUT, "score_KAT.npz"), PM=PM, RE=RE, CE=CE, users=users, items=items, cats=cats,
                 coef=np.float64(0.99), score_f64=oracle.inference_f64(PM, RE, CE, users, items, cats),
                 score_f32=oracle.inference_f32(PM, RE, CE, users, items, cats),
                 hand=np.array([3.4625]), hand_high=np.array([3.5]), hand_low=np.array([-0.25]))
    for case in CASES:
        name, U, I, C, E, B, seed = case[:7]
        coef = case[7] if len(case) > 7 else 0.99
        if keep(name):
            print(name, "kept")
            continue
        PM, RE, CE, users, items, cats = make_case(U, I, C, E, B, seed)
        np.savez(os.path.join(OUT, "score_%s.npz" % name), PM=PM, RE=RE, CE=CE, users=users, items=items,
                 cats=cats, coef=np.float64(coef),
                 score_f64=oracle.inference_f64(PM, RE, CE, users, items, cats, coef),
                 score_f32=oracle.inference_f32(PM, RE, CE, users, items, cats, coef))
        print(name, "ok")


if __name__ == "__main__":
    main()
