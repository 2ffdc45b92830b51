"""Retrieval's pruning bounds under cancellation (`m2d_topk_users`, option "topk_prune"; bounds in
csrc/m2d_catalogue.hip::grouped_pattern_terms).

A pruned call leaves out every mask pattern whose upper bound alpha_P + |w_P| max|r| is below a scan-start bound of the
user's k-th score (Model_Recommender.py:82-96 collapsed per pattern; the ranking itself is evaluate.py:63).  The bounds
have to hold for the scores the scan kernels COMPUTE, so their margins must follow what was summed, not what survived the
summation: with low-level rows that cancel (U_low[c] = -U_low[d] + eps noise) the f32 Gram matrix gives |w_P|^2 = 0 or a
negative number while the true |w_P| is not 0, and a margin relative to the result vanishes with it.  These tables are
built to sit in exactly those corners -- cancelling and repeated low-level rows (what Write_Memory produces for a dish of
several categories, Model_Recommender.py:111-119), high-level products that cancel from O(1) terms, equal alpha_P over
all patterns (identical category rows: only the low level ranks), low-level rows 1e-4 ... 1e-6 of the high level's scale
-- with >= 10^5 users per case, and every pruned form must return the plain scan's lists bit for bit ("topk_prune" = 0),
plus the float64 restatement's ranking on a sample (`_check`)."""
import numpy as np
import pytest

from helpers import COEFS
from test_gpu_catalogue import _check

pytestmark = pytest.mark.gpu

C = 4


def adversarial_tables(style, E, U, I, seed, eps=1e-4, low_scale=1.0, pats_per_dish=None):
    """Tables for one case.  Dishes spread over all 15 patterns (about I / 15 each), N(0, 1/E) rows unless the style says
    otherwise.  Returns PM, RE, CE, cats."""
    rng = np.random.default_rng(seed)
    s = 1.0 / np.sqrt(E)
    PM = (rng.standard_normal((U, C + 1, E)) * s).astype(np.float32)
    RE = (rng.standard_normal((I, E)) * s).astype(np.float32)
    CE = (rng.standard_normal((C, E)) * s).astype(np.float32)
    noise = (rng.standard_normal((U, 2, E)) * s).astype(np.float32)
    if style in ("anti", "anti_alpha0", "anti_equal_alpha"):
        # two cancelling pairs of low-level rows: patterns {0,1}, {2,3}, {0,1,2,3} have |w_P| = O(eps), the Gram sums O(1)
        PM[:, 2] = -PM[:, 1] + np.float32(eps) * noise[:, 0]
        PM[:, 4] = -PM[:, 3] + np.float32(eps) * noise[:, 1]
    if style == "anti_mixed":
        # per user one random pair cancels, another random pair repeats
        c = rng.integers(0, C, U); d = (c + rng.integers(1, C, U)) % C
        PM[np.arange(U), 1 + d] = -PM[np.arange(U), 1 + c] + np.float32(eps) * noise[:, 0]
        e = rng.integers(0, C, U); f = (e + rng.integers(1, C, U)) % C
        keep = (f != c) & (f != d)
        PM[np.flatnonzero(keep), 1 + f[keep]] = PM[np.flatnonzero(keep), 1 + e[keep]]
    if style == "same":
        PM[:, 2] = PM[:, 1]                     # U_low[1] = +U_low[0], U_low[3] = +U_low[2]
        PM[:, 4] = PM[:, 3]
    if style in ("anti_alpha0",):
        PM[:, 0] = 0.0                          # alpha_P = 0 for every pattern: the low level alone ranks
    if style in ("anti_equal_alpha", "equal_alpha"):
        CE[:] = CE[0]                           # identical category rows: alpha_P = a <U_high, CE_0> whatever the pattern
    if style == "hc_cancel":
        # <U_high, CE_c> ~ eps from terms of O(1): U_high = 8 (x - proj_CE0 x + eps CE_0), and CE_1 = -CE_0 + eps noise so that
        # patterns holding categories 0 and 1 cancel once more in the sum over c
        CE = (CE * 4).astype(np.float32)
        CE[1] = -CE[0] + np.float32(eps) * CE[1]
        x = PM[:, 0].astype(np.float64) * 8
        c0 = CE[0].astype(np.float64)
        x = x - np.outer(x @ c0 / (c0 @ c0), c0) + eps * c0
        PM[:, 0] = x.astype(np.float32)
    if style == "tiny_low":
        PM[:, 1:] *= np.float32(eps)            # low-level rows eps of the high-level scale: reach << rounding of alpha
    PM[:, 1:] *= np.float32(low_scale)
    pat = rng.integers(1, 2 ** C, I) if pats_per_dish is None else rng.choice(pats_per_dish, I)
    cats = ((pat[:, None] >> np.arange(C)[None, :]) & 1).astype(np.float32)
    return PM, RE, CE, cats


def lists_of_every_form(eng, users, k, forms=((0, 101), (1, 0), (1, 101), (1, 105), (2, 0), (4, 0))):
    """(prune option, forced split variant) -> (scores, ids); the first form is the plain scan over one dish range."""
    out = {}
    for prune, forced in forms:
        eng.set_option("topk_prune", prune)
        eng.set_option("variant", forced)
        s, i = eng.topk_users(users, k)
        eng.check()
        out[prune, forced] = (s.cpu().numpy(), i.cpu().numpy())
    eng.set_option("topk_prune", 1)
    eng.set_option("variant", 0)
    return out


def assert_forms_agree(out, what):
    base_key = next(iter(out))
    s0, i0 = out[base_key]
    for key, (s, i) in out.items():
        bad = np.flatnonzero((i != i0).any(1) | ~((s == s0) | (np.isnan(s) & np.isnan(s0))).all(1))
        assert bad.size == 0, (what, "form", key, "differs from", base_key, "for", bad.size, "users, first", int(bad[0]),
                               i[bad[0]], i0[bad[0]], s[bad[0]], s0[bad[0]])


CASES = [
    # style, E, x3 (split bf16 / exact f32), eps, low-level scale, dishes
    ("anti", 64, 1, 1e-3, 1.0, 7000), ("anti", 64, 1, 1e-4, 6.0, 7000), ("anti", 64, 1, 1e-5, 30.0, 7000), ("anti", 64, 1, 1e-6, 30.0, 7000),
    ("anti", 64, 0, 1e-4, 30.0, 7000), ("anti", 128, 1, 1e-4, 30.0, 5000), ("anti", 128, 0, 1e-5, 6.0, 5000),
    ("anti_alpha0", 64, 1, 1e-4, 1.0, 400), ("anti_alpha0", 64, 0, 1e-4, 30.0, 400), ("anti_alpha0", 128, 1, 1e-5, 6.0, 3000),
    ("anti_equal_alpha", 64, 1, 1e-4, 1.0, 400), ("anti_equal_alpha", 64, 0, 1e-3, 6.0, 3000), ("anti_equal_alpha", 128, 1, 1e-6, 30.0, 400),
    ("equal_alpha", 64, 1, 0.0, 1.0, 3000),
    ("anti_mixed", 64, 1, 1e-4, 30.0, 7000), ("anti_mixed", 64, 0, 1e-5, 6.0, 3000),
    ("same", 64, 1, 0.0, 1.0, 7000), ("same", 128, 0, 0.0, 30.0, 3000),
    ("hc_cancel", 64, 1, 1e-4, 1.0, 7000), ("hc_cancel", 64, 0, 1e-6, 1.0, 3000), ("hc_cancel", 128, 1, 1e-3, 0.05, 3000),
    ("tiny_low", 64, 1, 1e-4, 1.0, 7000), ("tiny_low", 64, 0, 1e-6, 1.0, 3000), ("tiny_low", 128, 1, 1e-5, 1.0, 3000), ("tiny_low", 64, 0, 1e-5, 1.0, 400),
]


# three of the styles again at the other blend coefficients (Train_recommender.py:61-62): the bounds' alpha_P scales with coef,
# their reach and rounding margins with |1 - coef| -- 0 leaves the cancelling low level alone to rank, 1 leaves none of it,
# 1.25 flips its sign
COEF_CASES = [c + (coef,) for coef in COEFS for c in (("anti", 64, 1, 1e-4, 6.0, 7000), ("anti_equal_alpha", 64, 0, 1e-3, 6.0, 3000),
                                                       ("hc_cancel", 64, 1, 1e-4, 1.0, 7000))]


# the split-bf16 cases again under the hi x hi first form of the pipelined kernel ("topk_form" 3; x3 = 3 below): the tiles'
# cross products are multiplied only where hi x hi comes within a bound of a threshold -- under cancellation the scores that
# matter are small beside |w| |r|, which is what that bound is made of
HI_FIRST_CASES = [(c[0], c[1], 3) + c[3:] + (0.99,) for c in CASES if c[2] == 1 and (c[3], c[4]) in ((1e-4, 1.0), (1e-4, 6.0), (1e-5, 30.0), (0.0, 1.0), (1e-4, 30.0),
                                                                                         (1e-5, 6.0), (1e-6, 30.0), (1e-3, 0.05), (1e-5, 1.0))] + \
                 [(c[0], c[1], 3) + c[3:] for c in COEF_CASES if c[2] == 1 and c[-1] in (0.0, 0.5, 1.25)]


@pytest.mark.parametrize("style,E,x3,eps,low_scale,I,coef", [c + (0.99,) for c in CASES] + COEF_CASES + HI_FIRST_CASES)
def test_pruned_lists_equal_the_plain_scan_under_cancellation(style, E, x3, eps, low_scale, I, coef):
    import torch
    from foodrec_amd import ScoringEngine
    U, k = 100_352, 10                                     # 392 blocks of 256 users
    hi_first, x3 = x3 == 3, min(x3, 1)
    seed = 9000 + sum(map(ord, style)) + E + 7 * x3 + I + int(-np.log10(eps)) if eps else 9000 + sum(map(ord, style)) + E + I
    PM, RE, CE, cats = adversarial_tables(style, E, U, I, seed, eps=eps, low_scale=low_scale)
    eng = ScoringEngine(PM, RE, CE, coef=coef)
    eng.set_dish_categories(cats)
    eng.set_option("topk_bf16x3", x3)
    eng.set_option("topk_form", 3 if hi_first else 0)
    users = torch.arange(U, dtype=torch.int32, device="cuda")
    out = lists_of_every_form(eng, users, k)
    assert eng.last_kernel() == (("m2d_topk_grouped_bf16x3" if x3 else "m2d_topk_grouped") if coef != 1.0 else "m2d_topk_high_level_only")
    assert_forms_agree(out, (style, E, x3, eps, low_scale, I))
    sample = np.random.default_rng(seed).choice(U, 24, replace=False)
    _check(eng, PM, RE, CE, cats, sample, k)
    eng.close()


@pytest.mark.parametrize("style,E,x3", [("anti_alpha0", 64, 1), ("anti_equal_alpha", 64, 0), ("anti", 128, 1)])
def test_small_groups_probed_whole(style, E, x3):
    """Catalogues whose patterns hold 10 to 40 dishes: the probe rows cover whole groups, so the scan-start bound IS the best
    pattern's k-th score less the margin -- the tightest bound the plan can produce, and the easiest to overshoot."""
    import torch
    from foodrec_amd import ScoringEngine
    U, k = 100_352, 10
    for I, eps, scale in ((180, 1e-4, 30.0), (330, 1e-5, 6.0), (600, 1e-3, 1.0)):
        PM, RE, CE, cats = adversarial_tables(style, E, U, I, seed=E + I, eps=eps, low_scale=scale)
        cats[:150] = ((np.arange(150)[:, None] % 15 + 1 >> np.arange(C)[None, :]) & 1).astype(np.float32)     # >= 10 dishes of every pattern
        eng = ScoringEngine(PM, RE, CE)
        eng.set_dish_categories(cats)
        eng.set_option("topk_bf16x3", x3)
        users = torch.arange(U, dtype=torch.int32, device="cuda")
        assert_forms_agree(lists_of_every_form(eng, users, k, forms=((0, 101), (1, 0), (1, 101), (2, 0))), (style, E, x3, I))
        _check(eng, PM, RE, CE, cats, np.arange(0, U, 9001), k)
        eng.close()


@pytest.mark.parametrize("style,E,x3,eps,low_scale,I", [
    ("anti_alpha0", 64, 1, 1e-4, 1.0, 400), ("anti_alpha0", 64, 1, 1e-3, 30.0, 2000), ("anti_alpha0", 64, 0, 1e-4, 6.0, 400),
    ("anti_alpha0", 128, 1, 1e-5, 30.0, 1000), ("anti_alpha0", 128, 0, 1e-4, 1.0, 400),
    ("anti_equal_alpha", 64, 1, 1e-4, 30.0, 400), ("anti_equal_alpha", 64, 0, 1e-3, 6.0, 1000), ("anti_equal_alpha", 128, 1, 1e-4, 30.0, 400),
    ("anti", 64, 1, 1e-4, 30.0, 400), ("anti", 64, 0, 1e-3, 30.0, 1000),
])
def test_catalogue_of_cancelling_patterns_only(style, E, x3, eps, low_scale, I):
    """Every dish carries categories {0,1}, {2,3} or all four -- the patterns whose low-level rows cancel -- so the whole
    ranking happens among scores of size eps, where a |w_P|^2 taken from the f32 Gram matrix is 0 or noise: with margins
    relative to that result a pattern's bounds collapse to alpha_P +- 1e-30 and another cancelling pattern's probe bound
    prunes it although its dishes score as high (the bounds of round 3 return other lists than the plain scan here)."""
    import torch
    from foodrec_amd import ScoringEngine
    U, k = 100_352, 10
    seed = 500 + sum(map(ord, style)) + E + 7 * x3 + I
    PM, RE, CE, cats = adversarial_tables(style, E, U, I, seed, eps=eps, low_scale=low_scale, pats_per_dish=[3, 12, 15])
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    eng.set_option("topk_bf16x3", x3)
    users = torch.arange(U, dtype=torch.int32, device="cuda")
    assert_forms_agree(lists_of_every_form(eng, users, k), (style, E, x3, eps, low_scale, I))
    _check(eng, PM, RE, CE, cats, np.random.default_rng(seed).choice(U, 24, replace=False), k)
    eng.close()
