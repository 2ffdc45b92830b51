"""Shared test helpers.  The oracle is the checker here and nowhere else."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star tolerance: scores match "within 1e-4 fp32"; stated as absolute-or-relative because
# |score| grows with E and table magnitude ("Reduction order").
TOL = 1e-4

# high_level_score_coefficient values the parity tests run beside the reference's default 0.99 (Train_recommender.py:61-62;
# `1 - coef` is taken in float32, Model_Recommender.py:17, :96): low level only, an even blend, a heavier low level than the
# default, high level only (every dish of a mask pattern then scores the same), and a NEGATIVE low-level weight.
COEFS = [0.0, 0.5, 0.9, 1.0, 1.25]


def assert_scores_close(got, ref, tol=TOL, what=""):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    nan_g, nan_r = np.isnan(got), np.isnan(ref)
    assert np.array_equal(nan_g, nan_r), "%s: NaN positions differ (%d vs %d)" % (what, nan_g.sum(), nan_r.sum())
    ok = ~nan_r
    err = np.abs(got[ok] - ref[ok])
    bound = tol * np.maximum(1.0, np.abs(ref[ok]))
    assert np.all(err <= bound), "%s: max err %.3e (bound %.1e)" % (what, err.max(), tol)
    return float(err.max()) if err.size else 0.0


def assert_scores_match_nonfinite(got, ref, tol=TOL, what=""):
    """assert_scores_close for results that may hold +-inf as well (tables with inf in them): NaN at the same pairs, the
    same infinities, the finite scores within the tolerance."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), "%s: NaN positions differ (%d vs %d)" % (what, np.isnan(got).sum(), np.isnan(ref).sum())
    inf = np.isinf(ref)
    assert np.array_equal(np.isinf(got), inf) and np.array_equal(got[inf], ref[inf]), "%s: infinities differ" % what
    fin = np.isfinite(ref)
    return assert_scores_close(got[fin], ref[fin], tol, what)


def score_cases():
    return sorted(glob.glob(os.path.join(GOLDEN, "score_*.npz")))


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def random_case(U, I, C, E, B, seed, zero_rows=True):
    rng = np.random.default_rng(seed)
    s = 1.0 / np.sqrt(E)
    PM = (rng.standard_normal((U, C + 1, E)) * s).astype(np.float32)
    RE = (rng.standard_normal((I, E)) * s).astype(np.float32)
    CE = (rng.standard_normal((C, E)) * s).astype(np.float32)
    users = rng.integers(0, U, B).astype(np.int32)
    items = rng.integers(0, I, B).astype(np.int32)
    cats = rng.integers(0, 2, (B, C)).astype(np.float32)
    if not zero_rows:
        cats[cats.sum(1) == 0, 0] = 1.0
    return PM, RE, CE, users, items, cats
