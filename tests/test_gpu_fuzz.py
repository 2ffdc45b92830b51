"""Property-based parity of the pair-score path on arbitrary shapes (hypothesis): any embed size (the vectorised kernels
need E % 4 == 0 and C == 4, everything else takes the generic kernel), any batch length, masks as arbitrary
non-negative weights with zero rows (0/0 -> NaN, Model_Recommender.py:79), against the float64 restatement; and the
latency / throughput forms of the kernel against each other, bit for bit."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from helpers import assert_scores_close

pytestmark = pytest.mark.gpu


@st.composite
def cases(draw):
    C = draw(st.sampled_from([4, 4, 4, 1, 2, 3, 5, 6]))
    E = draw(st.one_of(st.integers(1, 64).map(lambda x: 4 * x), st.integers(1, 300)))
    U = draw(st.integers(1, 40))
    I = draw(st.integers(1, 40))
    B = draw(st.integers(1, 300))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    weights = draw(st.booleans())
    return C, E, U, I, B, seed, weights


@settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(cases())
def test_any_shape_matches_the_restatement(case):
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    C, E, U, I, B, seed, weights = case
    rng = np.random.default_rng(seed)
    s = 1.0 / np.sqrt(E)
    PM = (rng.standard_normal((U, C + 1, E)) * s).astype(np.float32)
    RE = (rng.standard_normal((I, E)) * s).astype(np.float32)
    CE = (rng.standard_normal((C, E)) * s).astype(np.float32)
    users = rng.integers(0, U, B).astype(np.int32)
    items = rng.integers(0, I, B).astype(np.int32)
    cats = rng.integers(0, 2, (B, C)).astype(np.float32)            # zero rows included: NaN scores
    if weights:
        cats *= rng.choice([0.25, 1.0, 3.0], size=cats.shape).astype(np.float32)
    eng = ScoringEngine(PM, RE, CE)
    dev = lambda a: torch.as_tensor(a, device="cuda")
    got = eng.score_pairs(dev(users), dev(items), dev(cats)).cpu().numpy(); eng.check()
    ref = oracle.inference_f64(PM, RE, CE, users, items, cats)
    assert_scores_close(got, ref, what="C%d E%d B%d %s" % (C, E, B, eng.last_kernel()))
    vec = C == 4 and E % 4 == 0 and E <= 256
    assert eng.last_kernel() == ("m2d_score_pairs_c4_small" if vec else "m2d_score_pairs_generic")
    host = eng.score_pairs_host(users, items, cats)                  # the host-buffer entry point, same bits
    assert np.array_equal(host, got, equal_nan=True)
    if vec:
        eng.set_option("variant", 11)                                # the throughput form on the same batch
        big = eng.score_pairs(dev(users), dev(items), dev(cats)).cpu().numpy(); eng.check()
        assert eng.last_kernel() == "m2d_score_pairs_c4"
        assert np.array_equal(big, got, equal_nan=True)
    eng.close()


@st.composite
def retrieval_cases(draw):
    E = draw(st.sampled_from([32, 64, 128, 64, 128, 48, 200]))
    U = draw(st.integers(1, 70))
    I = draw(st.integers(1, 400))
    k = min(I, draw(st.sampled_from([1, 3, 10, 16, 17, 40])))        # the ABI asks for k <= min(64, I)
    n_users = draw(st.integers(1, min(U, 40)))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    n_nan = draw(st.integers(0, min(I, 4)))
    dup = draw(st.integers(0, I // 3))
    x3 = draw(st.booleans())
    splits = draw(st.sampled_from([0, 0, 102, 107, 164]))
    form = draw(st.sampled_from([0, 0, 1]))                             # split-bf16 kernel: pipelined (default) / first form
    return E, U, I, k, n_users, seed, n_nan, dup, x3, splits, form


@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(retrieval_cases())
def test_retrieval_any_catalogue(case):
    """m2d_topk_users on arbitrary catalogue sizes (fewer dishes than k, a single tile, ragged tails), with NaN dishes,
    exactly tied dishes, forced dish-range splits and both arithmetic forms: the checks of test_gpu_catalogue._check."""
    from foodrec_amd import ScoringEngine
    from test_gpu_catalogue import _check, _tables
    E, U, I, k, n_users, seed, n_nan, dup, x3, splits, form = case
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=seed, n_nan=n_nan, dup=dup)
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    eng.set_option("topk_bf16x3", int(x3))
    eng.set_option("topk_form", form)
    if splits:
        eng.set_option("variant", splits)
    users = np.random.default_rng(seed).choice(U, n_users, replace=False)
    _check(eng, PM, RE, CE, cats, users, k)
    eng.close()


@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 40), st.integers(1, 70), st.sampled_from([1, 5, 10, 20]),
       st.sampled_from([32, 64, 200]))
def test_segment_ranking_any_candidates(seed, nseg, L, K, E):
    """m2d_rank_candidates against the reference's dict + heapq.nlargest sequence (evaluate.py:53-63, restated in
    oracle.rank_candidates) on candidate lists full of repeated dishes and exactly tied scores, ragged lengths."""
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    rng = np.random.default_rng(seed)
    U, I, C = 30, 25, 4
    s = 1.0 / np.sqrt(E)
    PM = (rng.standard_normal((U, C + 1, E)) * s).astype(np.float32)
    RE = (rng.standard_normal((I, E)) * s).astype(np.float32)
    CE = (rng.standard_normal((C, E)) * s).astype(np.float32)
    cats = rng.integers(0, 2, (I, C)).astype(np.float32)
    cats[cats.sum(1) == 0, 0] = 1
    RE[I - 6:] = RE[:6]; cats[I - 6:] = cats[:6]                       # six pairs of dishes with bit-equal scores
    eng = ScoringEngine(PM, RE, CE); eng.set_dish_categories(cats)
    users = rng.integers(0, U, nseg).astype(np.int32)
    lens = rng.integers(1, L + 1, nseg).astype(np.int32)
    items = rng.integers(0, I, (nseg, L)).astype(np.int32)              # 25 dishes in up to 70 slots: many repeats
    dev = lambda a: torch.as_tensor(a, device="cuda")
    sc, ids, flags = eng.rank_candidates(dev(users), dev(items), K, lens=dev(lens)); eng.check()
    sc, ids, flags = sc.cpu().numpy(), ids.cpu().numpy(), flags.cpu().numpy()
    assert not flags.any()                                              # no NaN dish in this catalogue
    for r in range(nseg):
        cand = items[r, :lens[r]]
        scores = eng.score_pairs_bydish(dev(np.full(len(cand), users[r], np.int32)), dev(cand)).cpu().numpy()
        want = oracle.rank_candidates(cand.tolist(), scores.tolist(), K)
        got = [int(x) for x in ids[r] if x >= 0]
        assert got == want, (r, got, want)
        assert np.array_equal(sc[r, :len(want)], np.array([scores[np.where(cand == w)[0][-1]] for w in want], np.float32))
    eng.close()


@settings(max_examples=30, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(st.integers(0, 2 ** 31 - 1), st.sampled_from(["adam", "sgd", "adagrad", "rmsprop"]), st.integers(1, 6),
       st.one_of(st.integers(1, 40).map(lambda x: 4 * x), st.integers(1, 90)), st.integers(1, 200))
def test_training_step_any_shape(seed, learner, C, E, B):
    """One m2d_train_step on arbitrary (C, E, B) -- row widths that are not multiples of 4, batches smaller than a
    wave, every id repeated -- against the restatement (oracle/train_oracle.py, PARITY UNPINNED)."""
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import train_oracle as T
    rng = np.random.default_rng(seed)
    U, I = int(rng.integers(1, 30)), int(rng.integers(1, 30))
    s = 3.0 / np.sqrt(E)
    PM = (rng.standard_normal((U, C + 1, E)) * s).astype(np.float32)
    RE = (rng.standard_normal((I, E)) * s).astype(np.float32)
    CE = (rng.standard_normal((C, E)) * s).astype(np.float32)
    users = rng.integers(0, U, B).astype(np.int32)
    items = rng.integers(0, I, B).astype(np.int32)
    cats = rng.integers(0, 2, (B, C)).astype(np.float32)
    cats[cats.sum(1) == 0, 0] = 1.0
    labels = rng.integers(0, 2, B).astype(np.float32)
    lr = 0.01
    eng = ScoringEngine(PM.copy(), RE.copy(), CE.copy())
    eng.train_begin(learner, lr)
    st_ = T.TrainState(PM, RE, CE, learner, lr)
    ref_loss, ref_norm = st_.step(users, items, cats, labels)
    dev = lambda a: torch.as_tensor(a, device="cuda")
    out = eng.train_step(dev(users), dev(items), dev(cats), dev(labels)).cpu().numpy(); eng.check()
    assert abs(out[0] - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
    assert abs(out[1] - ref_norm) <= 1e-5 * max(1.0, ref_norm)
    tol = 1e-3 * lr if learner in ("adam", "rmsprop") else None
    for got, ref in ((eng.pm, st_.PM), (eng.re, st_.RE), (eng.ce, st_.CE)):
        err = np.abs(got.cpu().numpy().astype(np.float64) - ref)
        bound = tol if tol is not None else 1e-5 * np.maximum(1.0, np.abs(ref))
        assert np.all(err <= bound), (learner, C, E, B, err.max())
    eng.close()


@settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(st.sampled_from([32, 64, 64, 128, 24]), st.integers(1, 50), st.integers(1, 500), st.sampled_from([1, 4, 10, 16]),
       st.integers(0, 2 ** 31 - 1), st.booleans(), st.sampled_from([0, 0, 103]))
def test_retrieval_with_ingredients_any_catalogue(E, U, I, k, seed, weighted, splits):
    """m2d_topk_users with the ingredient table set (rows [H[d] | RE[d]] on the pattern-grouped kernel at E = 32 / 64,
    the dense kernel otherwise): returned scores against the float64 restatement, descending order, optimality, NaN
    dishes (empty ingredient list or empty mask) never ahead of a scored one."""
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    from test_gpu_catalogue import _tables
    k = min(k, I)
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=seed, n_nan=min(I // 5, 2))
    rng = np.random.default_rng(seed ^ 0x5A5A)
    R = int(rng.integers(1, 60))
    lens = rng.integers(0, 6, I)                                        # empty lists included
    off = np.zeros(I + 1, np.int32); off[1:] = np.cumsum(lens)
    ids = rng.integers(0, R, off[-1]).astype(np.int32)
    w = rng.uniform(0.5, 2.0, len(ids)).astype(np.float32) if weighted else None
    ING = (rng.standard_normal((R, E)) / np.sqrt(E)).astype(np.float32)
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    eng.set_ingredients(ING, off, ids, w)
    if splits:
        eng.set_option("variant", splits)
    users = np.arange(U, dtype=np.int32)
    s, idx = eng.topk_users(torch.as_tensor(users, device="cuda"), k); eng.check()
    grouped = E in (32, 64) and cats.sum() > 0
    assert (eng.last_kernel() == "m2d_topk_grouped_bf16x3") == grouped, eng.last_kernel()
    s, idx = s.cpu().numpy(), idx.cpu().numpy()
    for u in range(U):
        ref = oracle.inference_ingredients(PM, RE, ING, off, ids, w, np.full(I, u), np.arange(I), cats)
        assert len(set(idx[u].tolist())) == k and idx[u].min() >= 0 and idx[u].max() < I
        assert_scores_close(s[u], ref[idx[u]], what="user %d" % u)
        key = np.where(np.isnan(s[u]), -np.inf, s[u])
        assert np.all(key[:-1] >= key[1:])
        rest = np.delete(np.where(np.isnan(ref), -np.inf, ref), idx[u])
        if rest.size and np.isfinite(key[-1]):
            assert rest.max() <= key[-1] + 1e-4 * max(1.0, abs(key[-1]))
        if rest.size and not np.isfinite(key[-1]):
            assert not np.isfinite(rest).any()                          # a NaN made the list only when nothing scored was left
    eng.close()


@settings(max_examples=20, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(st.sampled_from([(4, 64), (4, 128), (5, 32), (2, 64), (4, 256)]), st.integers(1, 30_000), st.integers(1, 60),
       st.integers(1, 40), st.integers(0, 2 ** 31 - 1), st.booleans(), st.booleans())
def test_mlp_head_any_batch(shape, B, U, I, seed, weighted, skip):
    """The 3-layer head's producer / consumer kernel on arbitrary batches: any mix of dish mask patterns (weighted masks,
    dishes without categories -> NaN), few dishes (buckets of very different sizes, most of them padding), grouped by
    pattern or not, against the float64 restatement."""
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    C, E = shape
    K = (C + 1) * E
    rng = np.random.default_rng(seed)
    s = 1.0 / np.sqrt(E)
    PM = (rng.standard_normal((U, C + 1, E)) * s).astype(np.float32)
    RE = (rng.standard_normal((I, E)) * s).astype(np.float32)
    CE = (rng.standard_normal((C, E)) * s).astype(np.float32)
    users = rng.integers(0, U, B).astype(np.int32)
    items = rng.integers(0, I, B).astype(np.int32)
    dish_cats = rng.integers(0, 2, (I, C)).astype(np.float32)       # all-zero rows included: NaN scores
    if weighted:
        dish_cats *= rng.uniform(0.1, 3.0, (I, C)).astype(np.float32)
    head = ((rng.standard_normal((K, 256)) * 4 / np.sqrt(K)).astype(np.float32), (rng.standard_normal(256) * 0.1).astype(np.float32),
            (rng.standard_normal((256, 64)) / 4).astype(np.float32), (rng.standard_normal(64) * 0.1).astype(np.float32),
            (rng.standard_normal(64) / 2).astype(np.float32), 0.125)
    eng = ScoringEngine(PM, RE, CE); eng.set_dish_categories(dish_cats); eng.set_mlp_head(*head)
    eng.set_option("skip_masked", int(skip))
    got = eng.score_pairs_mlp(torch.as_tensor(users, device="cuda"), torch.as_tensor(items, device="cuda")); eng.check()
    assert eng.last_kernel() == "m2d_mlp_pc_bf16x3"
    ref = oracle.inference_mlp(PM, RE, CE, dish_cats, *head, users, items)
    assert_scores_close(got.cpu().numpy(), ref, what="C%d E%d B%d" % (C, E, B))
