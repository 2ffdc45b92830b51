"""Tables that hold inf / NaN (a diverged training run): the DEFAULT configuration returns what the reference graph
returns.  Model_Recommender.py:82 multiplies the Personal_Memory row of every category a dish does not have by 0, and
0 * inf = NaN, so a score is NaN wherever such a row is not finite.  The kernels that leave those rows out (option
skip_masked = 1, the pattern-grouped forms) read the engine's "a table value is not finite" word and stop doing so:
the word is set by the table scan queued at m2d_create / m2d_tables_updated and by the engine's own writers."""
import numpy as np
import pytest

from helpers import assert_scores_close, assert_scores_match_nonfinite, random_case

pytestmark = pytest.mark.gpu


def _masks(rng, B, C):
    m = (rng.integers(1, 2 ** C, B)[:, None] >> np.arange(C)[None, :] & 1).astype(np.float32)     # non-empty 0/1 masks
    m[::7] *= rng.uniform(0.5, 2.0, (len(m[::7]), C)).astype(np.float32)                          # some weighted
    return m


# c4 throughput / latency forms (full and partial lane groups), the C != 4 vectorised form, the generic kernel
@pytest.mark.parametrize("C,E,B", [(4, 64, 20000), (4, 64, 700), (4, 200, 9000), (4, 24, 300), (3, 16, 9000), (6, 32, 12000),
                                   (4, 7, 500), (9, 64, 300)])
@pytest.mark.parametrize("poison", [np.inf, -np.inf, np.nan])
def test_default_scores_equal_the_reference_graph_on_nonfinite_tables(C, E, B, poison):
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I = 400, 300
    PM, RE, CE, users, items, _ = random_case(U, I, C, E, B, seed=E + C)
    rng = np.random.default_rng(E)
    cats = _masks(rng, B, C)
    ut, it, ct = (torch.as_tensor(x, device="cuda") for x in (users, items, cats))
    eng = ScoringEngine(PM, RE, CE)
    clean = eng.score_pairs(ut, it, ct).cpu().numpy(); eng.check()
    eng.set_option("skip_masked", 0)
    assert np.array_equal(clean, eng.score_pairs(ut, it, ct).cpu().numpy())          # finite tables: same bits either way
    # poison low-level rows of a few users: every pair of theirs whose dish lacks that category is NaN in the graph
    PM2 = PM.copy()
    for u in rng.choice(U, 5, replace=False):
        PM2[u, 1 + rng.integers(0, C), rng.integers(0, E)] = poison
    ref = oracle.inference_f64(PM2, RE, CE, users, items, cats)
    assert np.isnan(ref).any() and not np.isnan(ref).all()
    eng2 = ScoringEngine(PM2, RE, CE)
    assert eng2.get_option("skip_masked") == 1                                       # the default configuration
    got = eng2.score_pairs(ut, it, ct).cpu().numpy(); eng2.check()
    assert_scores_match_nonfinite(got, ref, what="default options, non-finite Personal_Memory")
    hb = eng2.score_pairs_host(users[:51], items[:51], cats[:51])                   # the reference-shaped host call
    assert_scores_match_nonfinite(hb, ref[:51], what="host feed")


def test_in_place_edits_are_rescanned_after_tables_updated():
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, C, E, B = 300, 200, 4, 64, 20000
    PM, RE, CE, users, items, _ = random_case(U, I, C, E, B, seed=3)
    cats = _masks(np.random.default_rng(1), B, C)
    pmt = torch.as_tensor(PM, device="cuda")
    eng = ScoringEngine(pmt, RE, CE)                     # the engine borrows pmt
    ut, it, ct = (torch.as_tensor(x, device="cuda") for x in (users, items, cats))
    a = eng.score_pairs(ut, it, ct).cpu().numpy(); eng.check()
    assert_scores_close(a, oracle.inference_f64(PM, RE, CE, users, items, cats))
    pmt[17, 2, 5] = float("inf")
    eng.tables_updated()
    PM2 = PM.copy(); PM2[17, 2, 5] = np.inf
    assert_scores_match_nonfinite(eng.score_pairs(ut, it, ct).cpu().numpy(), oracle.inference_f64(PM2, RE, CE, users, items, cats))
    pmt[17, 2, 5] = 0.25                                 # repaired: the next scan clears the word, rows are skipped again
    eng.tables_updated()
    PM2[17, 2, 5] = 0.25
    got = eng.score_pairs(ut, it, ct).cpu().numpy(); eng.check()
    assert not np.isnan(got).any()
    assert_scores_close(got, oracle.inference_f64(PM2, RE, CE, users, items, cats))
    # Recipe_Embedding / Category_Embedding are part of the scan (the grouped MLP / retrieval forms depend on them)
    ret = torch.as_tensor(RE, device="cuda")
    eng3 = ScoringEngine(PM, ret, CE)
    eng3.score_pairs(ut, it, ct); eng3.check()
    ret[5, 3] = float("nan")
    eng3.tables_updated()
    RE2 = RE.copy(); RE2[5, 3] = np.nan
    assert_scores_close(eng3.score_pairs(ut, it, ct).cpu().numpy(), oracle.inference_f64(PM, RE2, CE, users, items, cats))


@pytest.mark.parametrize("learner", ["adam", "adagrad"])
def test_a_training_step_that_writes_nonfinite_values_sets_the_word(learner):
    """No tables_updated() is needed after the engine's own writers.  The divergence is planted in an optimizer slot (Adam's
    m, Adagrad's accumulator) so that one step writes a non-finite value into a low-level row of user 17 only."""
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, C, E, B = 200, 100, 4, 32, 9000
    PM, RE, CE, users, items, _ = random_case(U, I, C, E, B, seed=8)
    cats = _masks(np.random.default_rng(2), B, C)
    ut, it, ct = (torch.as_tensor(x, device="cuda") for x in (users, items, cats))
    pmt = torch.as_tensor(PM.copy(), device="cuda")
    eng = ScoringEngine(pmt, RE, CE)
    assert not np.isnan(eng.score_pairs(ut, it, ct).cpu().numpy()).any()       # scanned: finite
    eng.train_begin(learner, lr=0.001)
    slot = torch.zeros_like(pmt) if learner == "adam" else torch.full_like(pmt, 0.1)
    slot[17, 2, 5] = float("inf") if learner == "adam" else float("nan")
    eng.train_slot(0, 0, restore=slot)
    tu = torch.full((64,), 17, dtype=torch.int32, device="cuda")              # user 17 is in the batch (adagrad touches its rows only)
    eng.train_step(tu, it[:64], torch.ones((64, C), device="cuda"), torch.ones(64, device="cuda"))
    eng.check()
    pm_now = pmt.cpu().numpy()
    assert not np.isfinite(pm_now[17, 2, 5]) and np.isfinite(np.delete(pm_now[17].ravel(), 2 * E + 5)).all()
    ref = oracle.inference_f64(pm_now, eng.re.cpu().numpy(), eng.ce.cpu().numpy(), users, items, cats)
    lacking = (users == 17) & (cats[:, 1] == 0)
    assert lacking.any() and np.isnan(ref[lacking]).all()
    got = eng.score_pairs(ut, it, ct).cpu().numpy(); eng.check()
    assert_scores_match_nonfinite(got, ref, what="after a diverged %s step" % learner)


def test_write_memory_that_adds_inf_sets_the_word():
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, C, E, B, L = 200, 100, 4, 32, 9000, 7
    PM, RE, CE, users, items, _ = random_case(U, I, C, E, B, seed=8)
    items = (items % (I - 1)).astype(np.int32)                                 # dish I - 1 is written from, never scored
    RE = RE.copy(); RE[I - 1, 2] = 3e38                                        # finite, but 10 x it is not
    cats = _masks(np.random.default_rng(2), B, C)
    ut, it, ct = (torch.as_tensor(x, device="cuda") for x in (users, items, cats))
    pmt = torch.as_tensor(PM.copy(), device="cuda")
    eng = ScoringEngine(pmt, RE, CE)
    assert not np.isnan(eng.score_pairs(ut, it, ct).cpu().numpy()).any()       # the scan has run, the word is clear
    gm = torch.zeros((L, C + 1, E), device="cuda")
    wu = torch.arange(8, dtype=torch.int32, device="cuda")
    wi = torch.full((8,), I - 1, dtype=torch.int32, device="cuda")
    wc = torch.zeros((8, C), device="cuda"); wc[:, 0] = 1                      # category 0 only: row 1 of the user block
    lab = torch.zeros((8, L), device="cuda"); lab[:, 0] = 1
    eng.write_memory(wu, wi, wc, torch.full((8, 1), 10.0, device="cuda"), lab, gm, 1.0, 0.1, 0.1, write_pm=True, write_gm=False)
    eng.check()
    pm_now = pmt.cpu().numpy()
    assert np.isinf(pm_now[:8, 1, 2]).all() and np.isfinite(pm_now[:8, 2:]).all() and np.isfinite(pm_now[8:]).all()
    ref = oracle.inference_f64(pm_now, RE, CE, users, items, cats)
    lacking = (users < 8) & (cats[:, 0] == 0)
    assert lacking.any() and np.isnan(ref[lacking]).all()
    got = eng.score_pairs(ut, it, ct).cpu().numpy(); eng.check()
    assert_scores_match_nonfinite(got, ref, what="after Write_Memory added inf")


def test_scan_covers_the_last_values_of_a_table_whose_size_is_not_a_multiple_of_four():
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, C, E, B = 401, 301, 4, 7, 9000                                       # 14 035 / 2 107 floats: 3 beyond the last float4
    PM, RE, CE, users, items, _ = random_case(U, I, C, E, B, seed=12)
    cats = _masks(np.random.default_rng(3), B, C)
    PM[-1, -1, -1] = np.inf
    users[:50] = U - 1
    eng = ScoringEngine(PM, RE, CE)
    got = eng.score_pairs(*(torch.as_tensor(x, device="cuda") for x in (users, items, cats))).cpu().numpy(); eng.check()
    assert_scores_match_nonfinite(got, oracle.inference_f64(PM, RE, CE, users, items, cats))


@pytest.mark.parametrize("with_ingredients", [False, True])
def test_grouped_mlp_head_keeps_every_block_when_it_has_to(with_ingredients):
    """The pattern-grouped producer / consumer kernel (B >= 16384): a non-finite table value, and a dish whose weights
    sum to 0 (NaN blocks even where m_c = 0 -- with an ingredient table block 0 stays finite, so nothing else makes the
    score NaN), run every k-block: same NaN positions as the ungrouped kernel and the restatement."""
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, C, E, B = 300, 200, 4, 64, 20000
    PM, RE, CE, users, items, _ = random_case(U, I, C, E, B, seed=31)
    rng = np.random.default_rng(4)
    dish_cats = (rng.integers(1, 16, I)[:, None] >> np.arange(C)[None, :] & 1).astype(np.float32)
    dish_cats[3] = 0                                      # empty mask
    dish_cats[9] = [1, -1, 0, 0]                          # weights that sum to 0
    K = (C + 1) * E
    W1 = (rng.standard_normal((K, 256)) / np.sqrt(K)).astype(np.float32)
    b1 = (rng.standard_normal(256) * 0.1).astype(np.float32)
    W2 = (rng.standard_normal((256, 64)) / 16).astype(np.float32)
    b2 = (rng.standard_normal(64) * 0.1).astype(np.float32)
    w3 = (rng.standard_normal(64) / 8).astype(np.float32)
    ut, it = torch.as_tensor(users, device="cuda"), torch.as_tensor(items, device="cuda")

    def run(pm):
        eng = ScoringEngine(pm, RE, CE)
        eng.set_dish_categories(dish_cats)
        eng.set_mlp_head(W1, b1, W2, b2, w3, 0.25)
        if with_ingredients:
            R = 50
            lens = np.random.default_rng(5).integers(1, 6, I)
            off = np.zeros(I + 1, np.int32); off[1:] = np.cumsum(lens)
            eng.set_ingredients((np.random.default_rng(6).standard_normal((R, E)) / 8).astype(np.float32), off,
                                np.random.default_rng(7).integers(0, R, int(off[-1])).astype(np.int32))
        g = eng.score_pairs_mlp(ut, it).cpu().numpy(); eng.check()
        assert eng.last_kernel() == "m2d_mlp_pc_bf16x3"
        eng.set_option("mlp_form", 1)                      # the ungrouped every-wave-gathers kernel
        u = eng.score_pairs_mlp(ut, it).cpu().numpy(); eng.check()
        return g, u

    g, u = run(PM)
    assert np.array_equal(np.isnan(g), np.isnan(u))
    assert np.isnan(g[(items == 3) | (items == 9)]).all() and not np.isnan(g[(items != 3) & (items != 9)]).any()
    assert np.allclose(g[~np.isnan(g)], u[~np.isnan(u)], rtol=1e-4, atol=1e-4)
    if not with_ingredients:
        assert_scores_close(g, oracle.inference_mlp(PM, RE, CE, dish_cats, W1, b1, W2, b2, w3, 0.25, users, items))
    PM2 = PM.copy()
    PM2[users[0], 1 + int(np.flatnonzero(dish_cats[items[0]] == 0)[0]) if (dish_cats[items[0]] == 0).any() else 1, 2] = np.inf
    g2, u2 = run(PM2)
    assert np.array_equal(np.isnan(g2), np.isnan(u2)) and np.isnan(g2).sum() > np.isnan(g).sum()


def test_retrieval_takes_the_dense_kernel_on_nonfinite_tables():
    """w_P = sum of the pattern's U_low rows leaves out the 0 * U_low[c] products: with a non-finite table value the
    pattern-grouped retrieval kernels are not used (the dense kernel multiplies everything, like the pair path)."""
    import torch
    from foodrec_amd import ScoringEngine
    U, I, C, E, k = 200, 500, 4, 64, 10
    PM, RE, CE, *_ = random_case(U, I, C, E, 1, seed=5)
    rng = np.random.default_rng(9)
    dish_cats = (rng.integers(1, 16, I)[:, None] >> np.arange(C)[None, :] & 1).astype(np.float32)
    pmt = torch.as_tensor(PM, device="cuda")
    eng = ScoringEngine(pmt, RE, CE)
    eng.set_dish_categories(dish_cats)
    users = torch.arange(U, dtype=torch.int32, device="cuda")
    s0, i0 = eng.topk_users(users, k); eng.check()
    assert eng.last_kernel().startswith("m2d_topk_grouped")
    pmt[7, 2, 0] = float("inf")
    eng.tables_updated()
    s1, i1 = eng.topk_users(users, k); eng.check()
    assert eng.last_kernel() == "m2d_topk_mfma"
    s0, s1, i0, i1 = (x.cpu().numpy() for x in (s0, s1, i0, i1))
    keep = np.arange(U) != 7
    assert np.array_equal(i0[keep], i1[keep])             # the other users are untouched
    # user 7: every dish without category 1 scores NaN in the graph and can never enter a list
    ut = torch.full((I,), 7, dtype=torch.int32, device="cuda")
    it = torch.arange(I, dtype=torch.int32, device="cuda")
    pair = eng.score_pairs(ut, it, torch.as_tensor(dish_cats, device="cuda")).cpu().numpy(); eng.check()
    assert np.isnan(pair[dish_cats[:, 1] == 0]).all()
    finite = np.flatnonzero(~np.isnan(pair))
    best = finite[np.argsort(-pair[finite], kind="stable")][:k]
    assert set(i1[7][:min(k, len(best))]) == set(best)


def test_retrieval_after_write_memory_added_inf_takes_the_dense_kernel():
    """The sorted dish rows of the pattern-grouped retrieval survive a Write_Memory on Personal_Memory (it touches no dish
    row), but the word "a table value is inf / NaN" may have been set by it: the next m2d_topk_users reads it again, takes the
    dense kernel and keeps the dishes whose score is NaN in the graph (0 * inf, Model_Recommender.py:82) out of the lists --
    the same answer as the pair path (`score_pairs`) gives for those users."""
    import torch
    from foodrec_amd import ScoringEngine
    U, I, C, E, k, L = 200, 500, 4, 64, 10, 7
    PM, RE, CE, *_ = random_case(U, I, C, E, 1, seed=15)
    RE = RE.copy(); RE[I - 1, 2] = 3e38                                        # finite, but 10 x it is not
    rng = np.random.default_rng(19)
    dish_cats = (rng.integers(1, 16, I)[:, None] >> np.arange(C)[None, :] & 1).astype(np.float32)
    pmt = torch.as_tensor(PM.copy(), device="cuda")
    eng = ScoringEngine(pmt, RE, CE)
    eng.set_dish_categories(dish_cats)
    users = torch.arange(U, dtype=torch.int32, device="cuda")
    s0, i0 = eng.topk_users(users, k); eng.check()
    assert eng.last_kernel().startswith("m2d_topk_grouped")                   # the grouped tables are built, the word is clear
    gm = torch.zeros((L, C + 1, E), device="cuda")
    wu = torch.arange(8, dtype=torch.int32, device="cuda")
    wi = torch.full((8,), I - 1, dtype=torch.int32, device="cuda")
    wc = torch.zeros((8, C), device="cuda"); wc[:, 0] = 1                      # category 0 only: row 1 of the user block
    lab = torch.zeros((8, L), device="cuda"); lab[:, 0] = 1
    eng.write_memory(wu, wi, wc, torch.full((8, 1), 10.0, device="cuda"), lab, gm, 1.0, 0.1, 0.1, write_pm=True, write_gm=False)
    eng.check()
    assert np.isinf(pmt[:8, 1, 2].cpu().numpy()).all()
    s1, i1 = eng.topk_users(users, k); eng.check()
    assert eng.last_kernel() == "m2d_topk_mfma"
    s1, i1, s0, i0 = s1.cpu().numpy(), i1.cpu().numpy(), s0.cpu().numpy(), i0.cpu().numpy()
    # users that were not written to: the same lists (another kernel, another rounding: where two dishes change places their
    # scores are closer than the split-bf16 product's error)
    differ = i0[8:] != i1[8:]
    close = np.abs(s0[8:] - s1[8:]) <= 3e-5 * np.maximum(1.0, np.abs(s1[8:]))        # (dish I - 1 scores ~1e36 for everybody)
    assert differ.mean() < 0.02 and np.all(close)
    it = torch.arange(I, dtype=torch.int32, device="cuda")
    ct = torch.as_tensor(dish_cats, device="cuda")
    for u in range(8):
        pair = eng.score_pairs(torch.full((I,), u, dtype=torch.int32, device="cuda"), it, ct).cpu().numpy(); eng.check()
        assert np.isnan(pair[dish_cats[:, 0] == 0]).all()                      # dishes without category 0: 0 * inf
        ranked = np.flatnonzero(~np.isnan(pair))
        best = ranked[np.argsort(-pair[ranked], kind="stable")][:k]
        assert set(i1[u][:len(best)].tolist()) == set(best.tolist()), u
        assert not (set(i1[u][:len(best)].tolist()) & set(np.flatnonzero(dish_cats[:, 0] == 0).tolist()))
