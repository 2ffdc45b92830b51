"""GPU parity of full-catalogue retrieval (m2d_topk_users, fp32-MFMA kernel + generic kernel) with the
oracle: scores within 1e-4*max(1,|ref|) of the float64 restatement at the returned ids, descending
order with NaN last, optimality of the returned set, and lower-id-first on exact ties."""
import numpy as np
import pytest

from helpers import COEFS, TOL, assert_scores_close

pytestmark = pytest.mark.gpu


def _tables(U, I, C, E, seed, n_nan=0, dup=0):
    rng = np.random.default_rng(seed)
    s = 1.0 / np.sqrt(E)
    PM = (rng.standard_normal((U, C + 1, E)) * s).astype(np.float32)
    RE = (rng.standard_normal((I, E)) * s).astype(np.float32)
    CE = (rng.standard_normal((C, E)) * s).astype(np.float32)
    pat = rng.integers(1, 2 ** C, I)
    cats = ((pat[:, None] >> np.arange(C)[None, :]) & 1).astype(np.float32)
    if n_nan:
        cats[rng.choice(I, n_nan, replace=False)] = 0
    if dup:                                     # dishes I-dup.. are copies of dishes 0..dup-1 -> exact ties
        RE[I - dup:] = RE[:dup]
        cats[I - dup:] = cats[:dup]
    return PM, RE, CE, cats


def _explain_mismatches(PM, RE, CE, cats, users, sa, ia, sb, ib, bound=3e-5, coef=0.99):
    """Two kernels' lists for the same users: wherever they hold different dishes at a position, the two dishes' exact
    (float64) scores are closer than the kernels' rounding -- `bound` x max(1, |s|): 3e-5 is the split-bf16 product's
    measured error -- so either order is a correct ranking at that precision.  Returns the number of such positions."""
    from oracle import m2d_oracle as oracle
    I = RE.shape[0]
    n = 0
    for r in np.flatnonzero(np.any(ia != ib, axis=1)):
        ref = oracle.inference_f64(PM, RE, CE, np.full(I, users[r]), np.arange(I), cats, coef)
        for p in np.flatnonzero(ia[r] != ib[r]):
            ea, eb = ref[ia[r, p]], ref[ib[r, p]]
            assert abs(ea - eb) <= bound * max(1.0, abs(ea)), (int(users[r]), int(p), int(ia[r, p]), int(ib[r, p]), ea, eb)
            n += 1
    return n


def _check(eng, PM, RE, CE, cats, users, k, user_base=0, dup=0, coef=None):
    import torch
    from oracle import m2d_oracle as oracle
    coef = eng.coef if coef is None else coef            # the engine's high_level_score_coefficient (Model_Recommender.py:17)
    s, ids = eng.topk_users(torch.as_tensor(users + user_base, dtype=torch.int32, device="cuda"), k)
    eng.check()
    s, ids = s.cpu().numpy(), ids.cpu().numpy()
    I = RE.shape[0]
    all_items = np.arange(I)
    for r, u in enumerate(users):
        ref = oracle.inference_f64(PM, RE, CE, np.full(I, u), all_items, cats, coef)
        got_ids = ids[r]
        n_valid = min(k, I)
        assert np.all(got_ids[:n_valid] >= 0) and np.all(got_ids[:n_valid] < I), (r, got_ids)
        assert len(set(got_ids[:n_valid].tolist())) == n_valid, "duplicate dish in top-k"
        assert_scores_close(s[r, :n_valid], ref[got_ids[:n_valid]], what="user %d" % u)
        key = np.where(np.isnan(s[r, :n_valid]), -np.inf, s[r, :n_valid])
        assert np.all(key[:-1] >= key[1:]), "not descending / NaN not last"
        # optimality: nothing left out beats the k-th returned score by more than the tolerance
        rest = np.delete(np.where(np.isnan(ref), -np.inf, ref), got_ids[:n_valid])
        if rest.size and np.isfinite(key[n_valid - 1]):
            kth = key[n_valid - 1]
            assert rest.max() <= kth + TOL * max(1.0, abs(kth)), (u, rest.max(), kth)
        # exact ties (bit-equal scores) go to the lower dish id
        for a in range(n_valid - 1):
            if s[r, a] == s[r, a + 1]:
                assert got_ids[a] < got_ids[a + 1], (u, got_ids[a], got_ids[a + 1])
        # ... also where the tie decides what the list holds: dishes I - dup.. are copies of dishes 0..dup - 1 (same row,
        # same mask, same score to the bit), so a returned copy has its lower-id original in the list as well
        if dup:
            held = set(got_ids[:n_valid].tolist())
            for d in got_ids[:n_valid]:
                if d >= I - dup and not np.isnan(ref[d]):
                    assert int(d) - (I - dup) in held, (u, int(d))


def _assert_ids_are_the_oracles_where_clear(ids, PM, RE, CE, cats, users, k, coef, gap=1e-6, what=""):
    """Index parity with `heapq.nlargest` over the float64 restatement (oracle.topk_catalogue: score desc, ties to the lower
    dish id, evaluate.py:63): the id at every position whose float64 score is either bit-equal to a neighbour's (a structural
    tie: id order decides) or more than `gap` away from both neighbours' -- the (k + 1)-th best included -- is the oracle's.
    Returns the fraction of positions checked."""
    from oracle import m2d_oracle as oracle
    ref_s, ref_i = oracle.topk_catalogue(PM, RE, CE, cats, users, k + 1, coef)
    checked = 0
    for r in range(len(users)):
        rs = np.where(np.isnan(ref_s[r]), -np.inf, ref_s[r])
        with np.errstate(invalid="ignore"):
            before = np.concatenate([[np.inf], rs[:k - 1] - rs[1:k]])
            after = rs[:k] - rs[1:k + 1]
        before = np.where(np.isnan(before), 0.0, before)   # -inf - -inf: NaN dishes, in id order
        after = np.where(np.isnan(after), 0.0, after)
        ok = ((before > gap) | (before == 0)) & ((after > gap) | (after == 0))
        assert np.array_equal(ids[r][ok], ref_i[r, :k][ok]), (what, int(users[r]), ids[r], ref_i[r, :k], ref_s[r])
        checked += int(ok.sum())
    return checked / float(len(users) * k)


@pytest.mark.parametrize("E,C", [(32, 4), (64, 4), (128, 4), (200, 4), (16, 3), (20, 4), (100, 4), (48, 4), (256, 4), (260, 4), (22, 4)])
@pytest.mark.parametrize("k", [1, 10, 16, 17, 64])
def test_topk_users_shapes(E, C, k):
    from foodrec_amd import ScoringEngine
    U, I = 150, 333
    PM, RE, CE, cats = _tables(U, I, C, E, seed=E + k, n_nan=3, dup=20)
    coef = ([0.99] + COEFS)[(E // 4 + k) % 6]            # every kernel family meets every blend coefficient over the grid
    eng = ScoringEngine(PM, RE, CE, coef=coef)
    eng.set_dish_categories(cats)
    users = np.random.default_rng(1).integers(0, U, 45)
    _check(eng, PM, RE, CE, cats, users, k, dup=20)
    mfma = (C, E) in ((4, 32), (4, 64), (4, 128))
    padded = C == 4 and not mfma and E % 4 == 0 and E <= 256     # sorted dish rows zero-padded to 32 / 64 / 128 / 256 floats
    grouped = "m2d_topk_grouped_bf16x3" if E in (64, 128) else "m2d_topk_grouped"      # default: split-bf16
    if padded:
        want = "m2d_topk_grouped" if k <= 16 else "m2d_topk_generic"
    else:
        want = "m2d_topk_generic" if not mfma else (grouped if k <= 16 else "m2d_topk_mfma")
    direct = lambda name: "m2d_topk_high_level_only" if coef == 1.0 and name.startswith("m2d_topk_grouped") else name
    assert eng.last_kernel() == direct(want)                 # (coef = 1: the pattern-grouped path reads its lists off, no scan)
    if padded and k <= 16:
        eng.set_option("topk_grouped", 0)             # the one-block-per-user kernel on the same data
        _check(eng, PM, RE, CE, cats, users, k)
        assert eng.last_kernel() == "m2d_topk_generic"
    if mfma and k <= 16:
        eng.set_option("topk_bf16x3", 0)              # exact-f32 MFMA, pattern-grouped
        _check(eng, PM, RE, CE, cats, users, k)
        assert eng.last_kernel() == direct("m2d_topk_grouped")
        eng.set_option("variant", 7)                  # the dense (C+1)E contraction on the same data
        _check(eng, PM, RE, CE, cats, users, k)
        assert eng.last_kernel() == "m2d_topk_mfma"


@pytest.mark.parametrize("E", [200, 100, 24])
def test_topk_padded_rows_many_dishes(E):
    """Embedding sizes without a kernel of their own (the reference's default is 200): dish rows zero-padded to the next
    instantiated width.  More dishes than one stage holds, group tails, dish splits, weighted masks fall back."""
    from foodrec_amd import ScoringEngine
    U, I = 70, 5000
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=E, n_nan=5, dup=40)
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    users = np.arange(0, U, 3)
    for k in (10, 16):
        _check(eng, PM, RE, CE, cats, users, k)
        assert eng.last_kernel() == "m2d_topk_grouped"
    eng.set_dish_categories(cats * 0.5)
    _check(eng, PM, RE, CE, cats * 0.5, users, 10)
    assert eng.last_kernel() == "m2d_topk_generic"


def test_topk_dish_splits_and_tail_tiles():
    from foodrec_amd import ScoringEngine
    U, I, E = 300, 2500, 64
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=5, n_nan=4, dup=64)
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    users = np.arange(0, 290, 3)
    for forced in (101, 103, 108, 164, 228, 612):      # 1, 3, 8, 64 dish-range splits; 128 and 512 take the two-pass merge
        eng.set_option("variant", forced)
        _check(eng, PM, RE, CE, cats, users, 10)
        _check(eng, PM, RE, CE, cats, users[:1], 10)   # a single query
    eng.set_option("variant", 0)
    _check(eng, PM, RE, CE, cats, users, 10)           # automatic split choice
    eng.set_option("variant", 9)                       # generic kernel on the same data
    _check(eng, PM, RE, CE, cats, users[:8], 10)


def test_topk_fewer_dishes_than_k_and_all_nan():
    import torch
    from foodrec_amd import ScoringEngine
    PM, RE, CE, cats = _tables(40, 12, 4, 64, seed=9)
    cats[3] = 0
    eng = ScoringEngine(PM, RE, CE); eng.set_dish_categories(cats)
    with pytest.raises(ValueError):
        eng.topk_users(torch.zeros(2, dtype=torch.int32, device="cuda"), 13)      # k > I
    s, ids = eng.topk_users(torch.arange(5, dtype=torch.int32, device="cuda"), 12)
    eng.check()
    ids = ids.cpu().numpy(); s = s.cpu().numpy()
    assert np.all(np.sort(ids, axis=1) == np.arange(12)[None, :])
    assert np.all(ids[:, -1] == 3) and np.all(np.isnan(s[:, -1]))                 # NaN dish ranks last


def test_topk_user_shard_and_bad_ids():
    import torch
    from foodrec_amd import ScoringEngine
    PM, RE, CE, cats = _tables(100, 200, 4, 64, seed=10)
    eng = ScoringEngine(PM, RE, CE, user_base=5000); eng.set_dish_categories(cats)
    _check(eng, PM, RE, CE, cats, np.arange(0, 100, 7), 10, user_base=5000)
    with pytest.raises(IndexError):
        eng.topk_users(torch.tensor([5000, 42], dtype=torch.int32, device="cuda"), 5); eng.check()


def test_topk_matches_pair_kernel_scores():
    """The two device paths (factored MFMA vs fused pair kernel) agree on the same (user, dish)."""
    import torch
    from foodrec_amd import ScoringEngine
    PM, RE, CE, cats = _tables(500, 4000, 4, 128, seed=11)
    eng = ScoringEngine(PM, RE, CE); eng.set_dish_categories(cats)
    users = torch.arange(0, 500, dtype=torch.int32, device="cuda")
    s, ids = eng.topk_users(users, 10); eng.check()
    pair = eng.score_pairs_bydish(users.repeat_interleave(10), ids.reshape(-1).contiguous()); eng.check()
    assert_scores_close(s.reshape(-1).cpu().numpy(), pair.cpu().numpy())


def test_topk_weighted_masks_use_the_dense_kernel():
    """Masks that are not 0/1 (the placeholder is float: any weight is legal) cannot be pattern-grouped."""
    from foodrec_amd import ScoringEngine
    U, I, E = 100, 500, 64
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=21, n_nan=2)
    cats = cats * np.random.default_rng(3).uniform(0.5, 2.0, cats.shape).astype(np.float32)
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    _check(eng, PM, RE, CE, cats, np.arange(0, 100, 3), 10)
    assert eng.last_kernel() == "m2d_topk_mfma"


def test_topk_grouped_group_boundaries():
    """Few dishes per pattern (every group ends in a partial tile), a pattern with no dish, many empty masks."""
    from foodrec_amd import ScoringEngine
    rng = np.random.default_rng(8)
    U, I, E = 70, 150, 64
    PM, RE, CE, _ = _tables(U, I, 4, E, seed=22)
    pat = rng.choice([0, 1, 2, 3, 5, 8, 15], I)              # patterns 4,6,7,9.. unused; pattern 0 -> NaN
    cats = ((pat[:, None] >> np.arange(4)[None, :]) & 1).astype(np.float32)
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    for x3 in (1, 0):
        eng.set_option("topk_bf16x3", x3)
        for k in (3, 10, 16):
            _check(eng, PM, RE, CE, cats, np.arange(U), k)
            assert eng.last_kernel() == ("m2d_topk_grouped_bf16x3" if x3 else "m2d_topk_grouped")
    cats2 = np.zeros_like(cats); cats2[:5, 1] = 1             # 5 rankable dishes, the rest NaN
    eng.set_dish_categories(cats2)
    _check(eng, PM, RE, CE, cats2, np.arange(10), 10)


@pytest.mark.parametrize("E", [64, 128])
@pytest.mark.parametrize("k", [1, 10, 16])
def test_topk_split_bf16_variant(E, k):
    """The opt-in split-bf16 ("bf16x3") retrieval kernel obeys the same 1e-4 bar as the exact-f32 one."""
    import torch
    from foodrec_amd import ScoringEngine
    U, I = 200, 3000
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=E + k + 1, n_nan=5, dup=40)
    PM *= 3.0                                            # |score| up to a few units: the tolerance is relative there
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    assert eng.get_option("topk_bf16x3") == 1           # the default for retrieval
    users = np.random.default_rng(2).integers(0, U, 90)
    _check(eng, PM, RE, CE, cats, users, k)
    assert eng.last_kernel() == "m2d_topk_grouped_bf16x3"
    for forced in (101, 104):
        eng.set_option("variant", forced)
        _check(eng, PM, RE, CE, cats, users, k)
    # measured error of the split product against the exact kernel on the same (user, dish) pairs
    eng.set_option("variant", 0)
    s3, i3 = eng.topk_users(torch.arange(U, dtype=torch.int32, device="cuda"), k); eng.check()
    exact = eng.score_pairs_bydish(torch.arange(U, dtype=torch.int32, device="cuda").repeat_interleave(k),
                                   i3.reshape(-1).contiguous()); eng.check()
    err = (s3.reshape(-1) - exact).abs().max().item()
    assert err < 3e-5, err


@pytest.mark.parametrize("E", [64, 128, 32])
def test_cross_pattern_ties_and_the_dense_option(E):
    """A user whose Personal_Memory block is all zero scores every dish 0: one global tie, and heapq.nlargest's rule
    (evaluate.py:63) returns the lowest ids.  So does every kernel: the dense one scans in id order, the pattern-grouped
    ones notice the tie at their lists' boundary and re-rank that user in id order (include/m2d.h).  Users without such
    ties get the same lists from both, up to swaps of dishes whose scores are closer than the kernels' rounding."""
    import torch
    from foodrec_amd import ScoringEngine
    U, I, k = 70, 3000, 10
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=E + 3)
    PM[5] = 0.0
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    users = torch.arange(U, dtype=torch.int32, device="cuda")
    for x3 in (1, 0):
        eng.set_option("topk_bf16x3", x3)
        eng.set_option("topk_grouped", 1)
        sg, ig = eng.topk_users(users, k); eng.check()
        assert eng.last_kernel().startswith("m2d_topk_grouped")
        eng.set_option("topk_grouped", 0)
        sd, idn = eng.topk_users(users, k); eng.check()
        assert eng.last_kernel() == "m2d_topk_mfma"
        sg, ig, sd, idn = sg.cpu().numpy(), ig.cpu().numpy(), sd.cpu().numpy(), idn.cpu().numpy()
        assert np.all(sg[5] == 0) and np.all(sd[5] == 0)
        assert idn[5].tolist() == list(range(k))                                  # dense: lowest ids
        assert ig[5].tolist() == list(range(k))                                   # pattern-grouped: the same
        # same scores within the bar; every position where the id lists differ is a pair of dishes closer than the rounding
        assert np.all(np.abs(sg - sd) <= TOL * np.maximum(1.0, np.abs(sd)))
        _explain_mismatches(PM, RE, CE, cats, np.arange(U), sg, ig, sd, idn)


@pytest.mark.parametrize("E", [64, 128, 32, 200])
@pytest.mark.parametrize("forced", [0, 103, 228])          # automatic splits, 3 dish-range splits, 128 (two-pass merge)
def test_structural_ties_resolve_to_the_lower_id(E, forced):
    """User vectors that score whole groups of dishes identically, so that ties decide what the list holds:
      user 0: all-zero block -- every dish scores 0;
      user 1: zero low-level rows -- a dish's score is alpha_P, the same for every dish of a mask pattern, and the scan
              order inside a pattern (descending row norm) is not the id order;
      user 2: zero low-level row of category 0 only -- ties among the dishes whose only category is 0;
      users 3..: ordinary.
    Every kernel form returns heapq.nlargest's list: descending score, equal scores by ascending dish id."""
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, k = 40, 2600, 10
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=E + 17, n_nan=7)
    PM[0] = 0.0
    PM[1, 1:] = 0.0
    PM[2, 1] = 0.0
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    users = torch.arange(U, dtype=torch.int32, device="cuda")
    ref_s, ref_i = oracle.topk_catalogue(PM, RE, CE, cats, range(3), k)
    forms = [(0, 0)] if E in (32, 200) else [(1, 0), (1, 1), (0, 0), (1, 3)]   # (topk_bf16x3, topk_form; 3: hi x hi first)
    for x3, form in forms:
        eng.set_option("topk_bf16x3", x3); eng.set_option("topk_form", form); eng.set_option("variant", forced)
        s, ids = eng.topk_users(users, k); eng.check()
        assert eng.last_kernel().startswith("m2d_topk_grouped")
        s, ids = s.cpu().numpy(), ids.cpu().numpy()
        for u in (0, 1):                                     # every score is tied with many others: the list is exactly the oracle's
            assert ids[u].tolist() == ref_i[u].tolist(), (x3, form, u, ids[u], ref_i[u])
            assert_scores_close(s[u], ref_s[u])
        _check(eng, PM, RE, CE, cats, np.arange(U), k)


@pytest.mark.parametrize("variant", [0, 13])       # 13: all but two of the tied users take the one-block-per-user repair kernel
def test_every_user_tied(variant):
    """A zero-initialised Personal_Memory table: every user is re-ranked in id order (the slow path at its worst)."""
    import torch
    from foodrec_amd import ScoringEngine
    U, I, E, k = 600, 3000, 64, 10
    _, RE, CE, cats = _tables(U, I, 4, E, seed=2, n_nan=4)
    PM = np.zeros((U, 5, E), np.float32)
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    eng.set_option("variant", variant)
    s, ids = eng.topk_users(torch.arange(U, dtype=torch.int32, device="cuda"), k); eng.check()
    assert eng.last_kernel() == "m2d_topk_grouped_bf16x3" and eng.get_option("topk_repaired") == U
    want = np.flatnonzero(cats.sum(1) > 0)[:k]               # the lowest ids with a non-empty mask (empty masks score NaN)
    assert np.all(ids.cpu().numpy() == want[None, :]) and np.all(s.cpu().numpy() == 0)


@pytest.mark.parametrize("E", [32, 64, 128, 200])
def test_exact_f32_lists_equal_the_oracle_where_gaps_are_clear(E):
    """Exact-f32 pattern-grouped kernel ("topk_bf16x3" = 0): the id at every position whose score is more than 1e-6 away
    from both neighbours' (the (k + 1)-th best included) is the oracle's."""
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, k = 60, 4000, 10
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=E + 5, n_nan=5, dup=30)
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    eng.set_option("topk_bf16x3", 0)
    s, ids = eng.topk_users(torch.arange(U, dtype=torch.int32, device="cuda"), k); eng.check()
    assert eng.last_kernel() == "m2d_topk_grouped"
    ids = ids.cpu().numpy()
    ref_s, ref_i = oracle.topk_catalogue(PM, RE, CE, cats, range(U), k + 1)
    clear = 0
    for u in range(U):
        gap_before = np.concatenate([[np.inf], ref_s[u, :k - 1] - ref_s[u, 1:k]])
        gap_after = ref_s[u, :k] - ref_s[u, 1:k + 1]
        ok = (gap_before > 1e-6) & (gap_after > 1e-6)
        assert np.array_equal(ids[u][ok], ref_i[u, :k][ok]), (u, ids[u], ref_i[u])
        # duplicates (bit-equal scores, gap 0) resolve to the lower id: the oracle's order again
        tied = ~ok & (np.concatenate([[False], ref_s[u, :k - 1] == ref_s[u, 1:k]]) | (ref_s[u, :k] == ref_s[u, 1:k + 1]))
        assert np.array_equal(ids[u][tied], ref_i[u, :k][tied]), (u, ids[u], ref_i[u])
        clear += int(ok.sum())
    assert clear > 0.9 * U * k


@pytest.mark.parametrize("E,k", [(64, 10), (128, 10), (64, 16), (128, 13)])
def test_both_forms_of_the_split_bf16_kernel_agree(E, k):
    """Option topk_form: 1 = first form, 0 / 2 = pipelined form (the default).  Same contraction, same lists; the
    pipelined form adds alpha_P when a score enters a list instead of starting the accumulator from it, so scores may
    differ in the last bits."""
    import torch
    from foodrec_amd import ScoringEngine
    U, I = 300, 5000
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=7 * E + k, n_nan=3, dup=40)
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    users = torch.arange(U, dtype=torch.int32, device="cuda")
    out = {}
    for form in (1, 2):
        eng.set_option("topk_form", form)
        s, i = eng.topk_users(users, k); eng.check()
        assert eng.last_kernel() == "m2d_topk_grouped_bf16x3"
        out[form] = (s.cpu().numpy(), i.cpu().numpy())
    (s1, i1), (s2, i2) = out[1], out[2]
    assert np.all(np.abs(s1 - s2) <= 1e-5 * np.maximum(1.0, np.abs(s1)))
    _explain_mismatches(PM, RE, CE, cats, np.arange(U), s1, i1, s2, i2)     # differing positions: scores closer than the rounding
    eng.set_option("topk_form", 1)
    _check(eng, PM, RE, CE, cats, np.arange(0, U, 7), k)                # the first form on its own against the oracle


def test_progress_word_timeout_is_reported_and_the_engine_recovers():
    """The E = 64 hi x hi first form's waves wait for their workgroup's progress words before they refill tiles 6 and 7 of a stage; a
    wave that waited past the bound latches M2D_ERR_KERNEL_TIMEOUT (include/m2d.h: that call's lists are invalid) and goes on instead
    of hanging the launch.  "variant" = 15 sets the bound to zero polls: check() must raise and name the cause, and the next call must
    be clean and return the lists of an undisturbed engine."""
    import torch
    from foodrec_amd import ScoringEngine
    U, I, E, k = 2048, 12000, 64, 10
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=777)
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    users = torch.arange(U, dtype=torch.int32, device="cuda")
    eng.set_option("topk_form", 3); eng.set_option("topk_prune", 0)
    s0, i0 = eng.topk_users(users, k); eng.check()
    eng.set_option("variant", 15)
    eng.topk_users(users, k)
    with pytest.raises(RuntimeError, match="progress words"):
        eng.check()
    eng.set_option("variant", 0)
    s1, i1 = eng.topk_users(users, k); eng.check()                     # the latch is cleared, the engine works
    assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))


@pytest.mark.parametrize("E,k,low_scale,coef", [(64, 10, 1.0, 0.99), (64, 16, 1.0, 0.99), (64, 10, 6.0, 0.99), (64, 10, 0.05, 0.99), (64, 10, 1.0, 0.5),
                                                (64, 10, 1.0, 0.0), (64, 16, 1.0, 1.25), (64, 10, 1.0, 0.9), (128, 10, 1.0, 0.99), (128, 16, 1.0, 0.99),
                                                (128, 10, 6.0, 0.5), (128, 13, 0.05, 1.25), (128, 10, 1.0, 0.0)])
def test_hi_first_form_same_lists_in_every_launch_shape(E, k, low_scale, coef):
    """Large catalogues (here forced, "topk_form" = 3): the body multiplies the hi x hi product only and
    compares against the threshold less a bound of the two cross products; a tile that then still has a candidate gets them from
    its rows in LDS.  A score is hi x hi + (lo x hi + hi x lo): its own arithmetic, so every launch shape must agree with every
    other bit for bit (pruned or not, any split count, a pattern switch right behind a candidate tile, tiles 6 and 7 of a stage
    completed out of the previous stage's buffer; E = 128: tiles 2 and 3, behind the barrier at the head of step "sub 2"), the ids
    are the exact-f32 kernel's, and the three-product form's scores are within its rounding."""
    import torch
    from foodrec_amd import ScoringEngine
    U, I = 1500, 9000
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=E + 141 + k, n_nan=6, dup=50)
    PM[:, 1:] *= low_scale
    PM[7] = 0.0
    eng = ScoringEngine(PM, RE, CE, coef=coef)
    eng.set_dish_categories(cats)
    users = torch.as_tensor(np.random.default_rng(15).permutation(U).astype(np.int32), device="cuda")
    eng.set_option("topk_form", 3)
    out = {}
    for prune in (0, 1, 2, 4):
        eng.set_option("topk_prune", prune)
        for forced in (0, 101, 103, 108):
            eng.set_option("variant", forced)
            s, i = eng.topk_users(users, k); eng.check()
            assert eng.last_kernel() == "m2d_topk_grouped_bf16x3"
            assert eng.get_option("topk_block_users") == 256
            out[prune, forced] = (s.cpu().numpy(), i.cpu().numpy())
    base = out[0, 101]
    for key, (s, i) in out.items():
        assert np.array_equal(i, base[1]) and np.array_equal(s, base[0], equal_nan=True), key
    eng.set_option("variant", 0); eng.set_option("topk_prune", 1)
    _check(eng, PM, RE, CE, cats, users.cpu().numpy()[:80], k, dup=50)
    _check(eng, PM, RE, CE, cats, np.array([7, 3, 7]), k)
    eng.set_option("topk_form", 4)                            # the three-product form: the same ids, scores within the rounding
    s4, i4 = eng.topk_users(users, k); eng.check()
    s4, i4 = s4.cpu().numpy(), i4.cpu().numpy()
    assert np.array_equal(i4, base[1])
    fin = np.isfinite(s4)
    assert np.all(np.abs(s4[fin] - base[0][fin]) <= 2e-5 * np.maximum(1.0, np.abs(s4[fin])))
    if not (E == 128 and k > 10):                             # (that instantiation keeps no left-out scores: near-ties are not re-ranked)
        eng.set_option("topk_bf16x3", 0)                      # and the exact-f32 kernel's ids
        s0, i0 = eng.topk_users(users, k); eng.check()
        assert np.array_equal(i0.cpu().numpy(), base[1])


@pytest.mark.parametrize("E,low_scale,coef", [(64, 1.0, 0.99), (128, 1.0, 0.99), (64, 6.0, 0.99), (128, 0.05, 0.99)] +
                         [(64, 1.0, c) for c in COEFS] + [(128, 1.0, 0.5), (128, 1.0, 1.0), (128, 6.0, 1.25), (64, 0.05, 0.0)])
def test_pattern_pruning_changes_nothing_but_the_work(E, low_scale, coef):
    """The pipelined kernel takes its users sorted by the mask of patterns that can reach their top-k, and a block steps
    through those patterns' tiles only (bounds from Cauchy-Schwarz: include/m2d.h, option "topk_prune").  Same lists, bit
    for bit, as the scan of everything ("topk_prune" = 0) -- whatever the low-level rows' scale makes of the bounds -- and,
    with the reference's 0.99 : 0.01 blend, far fewer tiles."""
    import torch
    from foodrec_amd import ScoringEngine
    U, I, k = 1500, 9000, 10
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=E + 41, n_nan=6, dup=50)
    PM[:, 1:] *= low_scale
    PM[7] = 0.0                                           # every pattern ties: nothing can be pruned for this user
    # the bounds scale with the coefficient (alpha_P with coef, the reach with |1 - coef|): 0 leaves no alpha_P to prune by, 0.5
    # and 0.9 widen the reach against it, 1.0 makes every dish of a pattern score alpha_P (a tie per group, for every user),
    # 1.25 turns the low-level operand's sign
    eng = ScoringEngine(PM, RE, CE, coef=coef)
    eng.set_dish_categories(cats)
    users = torch.as_tensor(np.random.default_rng(5).permutation(U).astype(np.int32), device="cuda")
    out = {}
    for prune in (0, 1, 2, 4):                            # 2: thresholds only, 4: patterns only (A/B forms of the option)
        eng.set_option("topk_prune", prune)
        for forced in (0, 101, 105):                      # automatic splits, one split, five splits
            eng.set_option("variant", forced)
            s, i = eng.topk_users(users, k); eng.check()
            assert eng.last_kernel() == ("m2d_topk_grouped_bf16x3" if coef != 1.0 else "m2d_topk_high_level_only")
            out[prune, forced] = (s.cpu().numpy(), i.cpu().numpy(), eng.get_option("topk_tiles_scanned"), eng.get_option("topk_tiles_full"))
    base = out[0, 101]
    for key, (s, i, scanned, full) in out.items():
        assert np.array_equal(i, base[1]) and np.array_equal(s, base[0], equal_nan=True), key
    if coef != 1.0:                                       # (coef = 1: no scan at all, see test_high_level_only_blend)
        assert out[0, 101][2] >= out[0, 101][3]           # everything is stepped through without pruning
    if low_scale <= 1.0 and coef == 0.99:                 # (at 0.9 the low level reaches ten times further: 0.8 of the tiles)
        assert out[1, 101][2] < 0.5 * out[0, 101][2], (out[1, 101][2], out[0, 101][2])
    eng.set_option("variant", 0); eng.set_option("topk_prune", 1)
    _check(eng, PM, RE, CE, cats, users.cpu().numpy()[:60], k, dup=50)
    _check(eng, PM, RE, CE, cats, np.array([7, 3, 7]), k)       # a single block of users: no sort, the union of three masks
    frac = _assert_ids_are_the_oracles_where_clear(base[1][:80], PM, RE, CE, cats, users.cpu().numpy()[:80], k, coef, gap=3e-5,
                                                   what="coef %s" % coef)
    assert frac > 0.3 or low_scale < 1.0, frac             # (a low level scaled down: most gaps are under 3e-5)


@pytest.mark.parametrize("E,x3", [(64, 1), (64, 0), (128, 1), (200, 0), (32, 0)])
@pytest.mark.parametrize("nU", [37, 3000])
def test_high_level_only_blend(E, x3, nU):
    """--high_level_score_coefficient 1.0 (Train_recommender.py:61-62): `1 - coef` is an exact float32 zero
    (Model_Recommender.py:96), every dish of a mask pattern scores alpha_P[u], and `heapq.nlargest` (evaluate.py:63) returns
    the best pattern's LOWEST ids, then the next pattern's -- for every user.  Every kernel form returns exactly that list
    (the pattern-grouped path takes it from the patterns' first ids, no scan; the dense kernel scans in id order), with the
    NaN dishes of empty masks last when a user's patterns run out."""
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, k = 3000, 2600, 10
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=E + 23, n_nan=9)
    pat = (cats * (1 << np.arange(4))[None, :]).sum(1).astype(int)
    cats[np.flatnonzero(pat == 5)[3:]] = [1, 0, 0, 0]          # pattern {0, 2} keeps three dishes: lists that span patterns
    eng = ScoringEngine(PM, RE, CE, coef=1.0)
    eng.set_dish_categories(cats)
    eng.set_option("topk_bf16x3", x3)
    users_np = np.random.default_rng(E).permutation(U)[:nU].astype(np.int32)
    users = torch.as_tensor(users_np, device="cuda")
    ref_s, ref_i = oracle.topk_catalogue(PM, RE, CE, cats, users_np[:200], k, 1.0)
    outs = []
    for grouped, prune, forced in ((1, 1, 0), (1, 0, 101), (1, 1, 105), (0, 1, 0)):
        if grouped == 0 and E == 200:
            continue                                          # (no dense MFMA kernel for K = 1000: the one-block-per-user kernel, covered elsewhere)
        eng.set_option("topk_grouped", grouped); eng.set_option("topk_prune", prune); eng.set_option("variant", forced)
        s, i = eng.topk_users(users, k); eng.check()
        assert eng.last_kernel() == ("m2d_topk_high_level_only" if grouped else "m2d_topk_mfma")
        assert eng.get_option("topk_repaired") == 0 and eng.get_option("topk_tiles_scanned") == 0      # nothing is scanned, nobody repaired
        s, i = s.cpu().numpy(), i.cpu().numpy()
        frac = _assert_ids_are_the_oracles_where_clear(i[:200], PM, RE, CE, cats, users_np[:200], k, 1.0, gap=3e-6, what=(grouped, prune, forced))
        assert frac > 0.95, frac                              # ties everywhere; only two patterns' alpha within 3e-6 are left out
        assert_scores_close(s[:200], ref_s)
        outs.append((s, i))
    for s, i in outs[1:3]:                                    # the pattern-grouped forms: the same bits
        assert np.array_equal(i, outs[0][1]) and np.array_equal(s.view(np.int32), outs[0][0].view(np.int32))
    eng.set_option("topk_grouped", 1); eng.set_option("topk_prune", 1); eng.set_option("variant", 0)
    _check(eng, PM, RE, CE, cats, users_np[:40], k)
    s16, i16 = eng.topk_users(users[:64], 16); eng.check()    # k = 16, and a catalogue with fewer rankable dishes than k
    _assert_ids_are_the_oracles_where_clear(i16.cpu().numpy(), PM, RE, CE, cats, users_np[:64], 16, 1.0, gap=3e-6)
    with pytest.raises(IndexError):                           # an id outside the shard is latched here as in the scan kernels
        eng.topk_users(torch.tensor([3, U + 7], dtype=torch.int32, device="cuda"), k); eng.check()
    few = np.zeros_like(cats); few[[5, 17, 900], 1] = 1; few[[3, 40], 2] = 1
    eng.set_dish_categories(few)
    _check(eng, PM, RE, CE, few, users_np[:20], k)
    s5, i5 = eng.topk_users(users[:20], k); eng.check()
    _assert_ids_are_the_oracles_where_clear(i5.cpu().numpy(), PM, RE, CE, few, users_np[:20], k, 1.0, gap=3e-6)


@pytest.mark.parametrize("E,x3", [(64, 1), (64, 0), (128, 1), (200, 0), (32, 0)])
def test_launch_order_and_the_exact_kernels_plan(E, x3):
    """A pruned launch with more (user block, dish range) items than CUs hands them out longest first (m2d_plan_items_*;
    "topk_prune" = 5 keeps the grid's order), and the exact-f32 kernel -- also the zero-padded one that serves the
    reference's embed_size 200 -- takes the same plan as the split-bf16 one.  The order of the work changes no list."""
    import torch
    from foodrec_amd import ScoringEngine
    U, I, k = 12000, 9000, 10
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=E + 77 + x3, n_nan=5, dup=40)
    PM[11] = 0.0
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    eng.set_option("topk_bf16x3", x3)
    users = torch.as_tensor(np.random.default_rng(9).permutation(U).astype(np.int32), device="cuda")
    out = {}
    for prune in (0, 5, 1):
        eng.set_option("topk_prune", prune)
        s, i = eng.topk_users(users, k); eng.check()
        assert eng.last_kernel() == ("m2d_topk_grouped_bf16x3" if x3 else "m2d_topk_grouped")
        out[prune] = (s.cpu().numpy(), i.cpu().numpy(), eng.get_option("topk_tiles_scanned"), eng.get_option("topk_tiles_full"))
    for prune in (5, 1):
        assert np.array_equal(out[prune][1], out[0][1]) and np.array_equal(out[prune][0], out[0][0], equal_nan=True), prune
    assert out[0][2] >= out[0][3] and out[1][2] == out[5][2] and out[1][2] < 0.6 * out[0][2], [o[2:] for o in out.values()]
    _check(eng, PM, RE, CE, cats, users.cpu().numpy()[:80], k, dup=40)


@pytest.mark.parametrize("E,x3,nU,I", [(200, 0, 1723, 27000), (200, 0, 1400, 27000), (128, 1, 1723, 40000), (64, 1, 1100, 27000)])
def test_more_than_64_dish_ranges_come_in_whole_groups(E, x3, nU, I):
    """Six or seven user blocks over a catalogue large enough for 512 / blocks = 85 / 73 dish ranges: the two-pass merge of
    more than 64 partial lists takes whole groups of 64, so the launch runs 64 (a randomized soak found 73 ranges shifting
    every user's list by one row); five blocks take 102 -> 64 by the serving rule.  (Stages of 2 / 4 tiles at E = 200 / 128
    leave room for that many ranges of at least four stages in these catalogues.)  Same lists as the plain scan."""
    import torch
    from foodrec_amd import ScoringEngine
    U, k = 3000, 10
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=E + nU, n_nan=3, dup=20)
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    eng.set_option("topk_bf16x3", x3)
    users = torch.as_tensor(np.random.default_rng(nU).permutation(U)[:nU].astype(np.int32), device="cuda")
    eng.set_option("topk_prune", 0); eng.set_option("variant", 101)
    s0, i0 = eng.topk_users(users, k); eng.check()
    eng.set_option("topk_prune", 1); eng.set_option("variant", 0)
    s1, i1 = eng.topk_users(users, k); eng.check()
    assert torch.equal(i0, i1) and torch.equal(s0.nan_to_num(nan=-7.0), s1.nan_to_num(nan=-7.0))
    _check(eng, PM, RE, CE, cats, users.cpu().numpy()[:40], k, dup=20)


@pytest.mark.parametrize("E,x3", [(64, 1), (128, 1), (64, 0), (200, 0)])
def test_re_ranked_scores_do_not_depend_on_where_a_dish_was_scored(E, x3):
    """The tie repair scans the pattern-sorted table, the groups of its listed users' relevant patterns only ("topk_prune" = 9:
    every group), so which block, lane group and in-flight slot scores a dish depends on who else is listed in the same pass.
    The score must not: hipcc fused the blend's second product into its add in one of the two in-flight copies (one rounding
    fewer; seen as one score of a re-ranked user one ulp apart between option forms).  Coarse tables: hundreds of listed users."""
    import torch
    from foodrec_amd import ScoringEngine
    U, I, k = 3000, 6000, 10
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=E + 5, dup=300)
    PM, RE, CE = (np.round(PM * 16) / 16).astype(np.float32), (np.round(RE * 16) / 16).astype(np.float32), (np.round(CE * 16) / 16).astype(np.float32)
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    eng.set_option("topk_bf16x3", x3)
    users = torch.as_tensor(np.random.default_rng(E).permutation(U).astype(np.int32), device="cuda")
    for refine in (0, 1):                                 # 0: every tied user goes through the repair; 1: only those with three or
        eng.set_option("topk_refine", refine)             # more dishes that close (the others are settled by m2d_topk_refine)
        out = {}
        for prune, forced in ((0, 101), (1, 0), (9, 0), (1, 103), (9, 107), (1, 0)):
            eng.set_option("topk_prune", prune); eng.set_option("variant", forced)
            s, i = eng.topk_users(users, k); eng.check()
            out.setdefault((prune, forced), []).append((s.cpu().numpy(), i.cpu().numpy(),
                                                        eng.get_option("topk_repaired") + (eng.get_option("topk_refined") if refine else 0)))
        s0, i0, rep = out[0, 101][0]
        assert rep >= 100, rep                             # the tables do tie many users' k-th score (re-ranked by the repair, or refined)
        for key, runs in out.items():
            for s, i, r in runs:
                assert np.array_equal(i, i0) and np.array_equal(s.view(np.int32), s0.view(np.int32)), (refine, key, r, rep)
    eng.set_option("topk_prune", 1); eng.set_option("variant", 0)
    _check(eng, PM, RE, CE, cats, users.cpu().numpy()[:40], k, dup=300)


@pytest.mark.parametrize("k", [10, 16])
def test_blocks_of_128_users_return_the_same_lists(k):
    """Pruned launches of the pipelined kernel over catalogues up to 8 192 tiles run blocks of 128 users (four waves, half-size
    stages, two blocks per CU; option "topk_block" forces 128 / 256): another launch shape, the same lists bit for bit -- also
    with a last block of 17 users, with dish ranges forced, and against the plain scan."""
    import torch
    from foodrec_amd import ScoringEngine
    U, I, E = 20000, 9000, 64
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=k + 300, n_nan=4, dup=60)
    PM[5] = 0.0
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    users = torch.as_tensor(np.random.default_rng(k).permutation(U)[:128 * 140 + 17].astype(np.int32), device="cuda")
    out = {}
    for prune, block, forced in ((0, 0, 101), (1, 0, 0), (1, 128, 0), (1, 256, 0), (1, 128, 105), (0, 128, 0), (7, 128, 0)):
        eng.set_option("topk_prune", prune); eng.set_option("topk_block", block); eng.set_option("variant", forced)
        s, i = eng.topk_users(users, k); eng.check()
        out[prune, block, forced] = (s.cpu().numpy(), i.cpu().numpy(), eng.get_option("topk_block_users"))
    assert out[1, 0, 0][2] == 128 and out[1, 256, 0][2] == 256 and out[1, 128, 105][2] == 128 and out[0, 0, 101][2] == 256
    s0, i0, _ = out[0, 0, 101]
    for key, (s, i, _) in out.items():
        assert np.array_equal(i, i0) and np.array_equal(s.view(np.int32), s0.view(np.int32)), key
    eng.set_option("topk_prune", 1); eng.set_option("topk_block", 0); eng.set_option("variant", 0)
    _check(eng, PM, RE, CE, cats, users.cpu().numpy()[:50], k, dup=60)


@pytest.mark.parametrize("nblocks", [800, 1600, 3100])
def test_many_user_blocks_take_fewer_dish_ranges(nblocks):
    """A pruned launch of blocks of 128 users is cut into 8 dish ranges up to 767 blocks, then 4 / 2 / 1 (about 4 096 items
    balance the launch; more only add prologues and merge work): the same lists as the plain scan, bit for bit, at each count --
    one range means no merge pass and no shared thresholds."""
    import torch
    from foodrec_amd import ScoringEngine
    U, I, E, k = 128 * nblocks + 40, 9000, 64, 10
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=nblocks, n_nan=3, dup=40)
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(cats)
    users = torch.as_tensor(np.random.default_rng(nblocks).permutation(U).astype(np.int32), device="cuda")
    eng.set_option("topk_prune", 0)
    s0, i0 = eng.topk_users(users, k); eng.check()
    eng.set_option("topk_prune", 1)
    s1, i1 = eng.topk_users(users, k); eng.check()
    assert eng.get_option("topk_block_users") == 128
    assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
    _check(eng, PM, RE, CE, cats, users.cpu().numpy()[:30], k, dup=40)


@pytest.mark.parametrize("E,k,I,coef", [(64, 10, 9000, 0.99), (128, 10, 7000, 0.99), (64, 16, 9000, 0.99), (64, 1, 3000, 0.99), (128, 5, 20000, 0.99)] +
                         [(64, 10, 9000, c) for c in COEFS] + [(128, 10, 7000, 0.5), (128, 10, 7000, 1.25), (64, 16, 3000, 0.0)])
def test_split_bf16_lists_are_the_exact_f32_kernels_lists(E, k, I, coef):
    """Dish ids are index output.  The split-bf16 kernel's products differ from the exact-f32 kernel's by up to ~1e-5 of
    |w||r|, which used to reorder dishes whose scores sit closer than that (about 0.2 % of the lists).  Both kernels now finish
    near-tied lists -- neighbouring scores, or the last entry and the best score left out, within 2 delta of each other -- in
    ONE arithmetic, the tie repair's plain f32 (m2d_topk_refine), so they return the same ids; "topk_refine" = 0 is the old
    behaviour.  Coarse-ish tables make near-ties common."""
    import torch
    from foodrec_amd import ScoringEngine
    U = 24000
    PM, RE, CE, cats = _tables(U, I, 4, E, seed=E + k + 900, n_nan=5, dup=30)
    eng = ScoringEngine(PM, RE, CE, coef=coef)            # delta_u (the near-tie margin) scales with |1 - coef| reach and |alpha|
    eng.set_dish_categories(cats)
    users = torch.as_tensor(np.random.default_rng(k).permutation(U).astype(np.int32), device="cuda")
    res = {}
    for x3 in (1, 3, 0):                                    # 3: split bf16, the hi x hi first form ("topk_form" 3; its own arithmetic)
        if x3 == 3 and E not in (64, 128):
            continue
        eng.set_option("topk_bf16x3", 1 if x3 else 0); eng.set_option("topk_form", 3 if x3 == 3 else 4)
        for prune, forced in ((1, 0), (0, 101), (1, 105)):
            eng.set_option("topk_prune", prune); eng.set_option("variant", forced)
            s, i = eng.topk_users(users, k); eng.check()
            res[x3, prune, forced] = (s.cpu().numpy(), i.cpu().numpy(), eng.get_option("topk_refined"))
    i_ref = res[0, 0, 101][1]
    for key, (s, i, refined) in res.items():
        assert np.array_equal(i, i_ref), (key, int((i != i_ref).any(1).sum()), refined)
    if coef != 1.0:
        assert res[1, 1, 0][2] > 0                          # some lists were near-tied and went through the refinement
    for x3 in (1, 3, 0):                                    # and inside one kernel every option form returns the same bits
        if (x3, 0, 101) not in res:
            continue
        base = res[x3, 0, 101][0]
        for key, (s, i, refined) in res.items():
            if key[0] == x3:
                assert np.array_equal(s.view(np.int32), base.view(np.int32)), key
    eng.set_option("topk_form", 0)
    eng.set_option("topk_bf16x3", 1); eng.set_option("topk_prune", 1); eng.set_option("variant", 0)
    eng.set_option("topk_refine", 0)                        # without it the two kernels disagree on some near-tie (what was measured)
    s_off, i_off = eng.topk_users(users, k); eng.check()
    eng.set_option("topk_refine", 1)
    _check(eng, PM, RE, CE, cats, users.cpu().numpy()[:60], k, dup=30)
