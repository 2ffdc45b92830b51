"""GPU parity of the build-defined 3-layer scoring head (m2d_set_mlp_head / m2d_score_pairs_mlp) against
the build's own float64 restatement.  The reference has no MLP: this pins only
the documented extension and its reduction to the reference score when the head contributes nothing."""
import numpy as np
import pytest

from helpers import COEFS, assert_scores_close, random_case

pytestmark = pytest.mark.gpu


def _head(K, H1, H2, rng, scale=1.0):
    W1 = (rng.standard_normal((K, H1)) * scale / np.sqrt(K)).astype(np.float32)
    b1 = (rng.standard_normal(H1) * 0.1).astype(np.float32)
    W2 = (rng.standard_normal((H1, H2)) * scale / np.sqrt(H1)).astype(np.float32)
    b2 = (rng.standard_normal(H2) * 0.1).astype(np.float32)
    w3 = (rng.standard_normal(H2) * scale / np.sqrt(H2)).astype(np.float32)
    return W1, b1, W2, b2, w3, 0.25


_MLP_SHAPES = [(128, 4, 256, 64, "m2d_mlp_mfma"), (64, 4, 256, 64, "m2d_mlp_mfma"), (256, 4, 256, 64, "m2d_mlp_mfma"),
               (32, 5, 256, 64, "m2d_mlp_mfma"),
               # K = (C + 1) E not a multiple of 64 (200 is the reference's default embed_size): zero-padded chunks
               (200, 4, 256, 64, "padded"), (100, 4, 256, 64, "padded"), (40, 4, 256, 64, "padded"), (240, 4, 256, 64, "padded"),
               (36, 3, 256, 64, "padded"),
               (64, 4, 128, 32, "m2d_mlp_generic"), (6, 3, 10, 7, "m2d_mlp_generic"), (30, 4, 256, 64, "m2d_mlp_generic")]
# split-bf16 MFMA: producer / consumer kernel (default), every-wave-gathers kernel; exact-f32 MFMA.  The padded form
# exists for the every-wave-gathers kernel only (either arithmetic), the generic kernel has one form.
_MLP_FORMS = {"m2d_mlp_mfma": [(1, 0), (1, 1), (0, 0)], "padded": [(1, 0), (0, 0)], "m2d_mlp_generic": [(1, 0)]}
_MLP_CASES = [shape + form for shape in _MLP_SHAPES for form in _MLP_FORMS[shape[4]]]


@pytest.mark.parametrize("E,C,H1,H2,kernel,x3,form", _MLP_CASES)
def test_mlp_scores_match_restatement(E, C, H1, H2, kernel, x3, form):
    """One test per (shape, arithmetic, kernel form); the batch sizes -- 1, around one and two tiles of 128 pairs, 3000 -- are
    looped inside (8 cases each, the blend coefficient varying with them)."""
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I = 300, 200
    for B in (1, 127, 128, 129, 255, 256, 257, 3000):
        PM, RE, CE, users, items, _ = random_case(U, I, C, E, B, seed=E + B)
        rng = np.random.default_rng(E + H1)
        dish_cats = rng.integers(0, 2, (I, C)).astype(np.float32)
        dish_cats[dish_cats.sum(1) == 0, 0] = 1
        dish_cats[3] = 0                                      # NaN dish
        K = (C + 1) * E
        head = _head(K, H1, H2, rng, scale=4.0)              # large enough that the head matters
        coef = ([0.99] + COEFS)[(E // 2 + B) % 6]            # the blend coefficient enters the head through z = PM[u] * Dt[d]
        eng = ScoringEngine(PM, RE, CE, coef=coef)
        ut, it = torch.as_tensor(users, device="cuda"), torch.as_tensor(items, device="cuda")
        with pytest.raises(ValueError):
            eng.score_pairs_mlp(ut, it)
        eng.set_dish_categories(dish_cats)
        eng.set_mlp_head(*head)
        eng.set_option("mlp_bf16x3", x3)
        eng.set_option("mlp_form", form)
        got = eng.score_pairs_mlp(ut, it); eng.check()
        if kernel == "padded":
            want = "m2d_mlp_mfma_bf16x3" if x3 else "m2d_mlp_mfma"
        else:
            want = kernel if kernel == "m2d_mlp_generic" or not x3 else ("m2d_mlp_mfma_bf16x3" if form else "m2d_mlp_pc_bf16x3")
        assert eng.last_kernel() == want
        ref = oracle.inference_mlp(PM, RE, CE, dish_cats, *head, users, items, coef=coef)
        base = oracle.inference_f64(PM, RE, CE, users, items, dish_cats[items], coef)
        ok = ~np.isnan(ref)
        if ok.sum() > 10:
            assert np.abs(ref[ok] - base[ok]).mean() > 1e-2, "head too small to be tested"
        assert_scores_close(got.cpu().numpy(), ref, what="E%d B%d" % (E, B))
        eng.close()


@pytest.mark.parametrize("E", [64, 128])
def test_mlp_large_table_instantiation_gives_the_same_bits(E):
    """m2d_mlp_pc<KCH, OFF32>: tables under 4 GiB take 32-bit byte offsets and scalar-base row loads, larger ones (up to 64 GiB)
    offsets in units of 16 B -- the same loads, the same arithmetic.  "variant" = 16 runs the second form on small tables: the same
    scores bit for bit, a bad id reported alike."""
    import torch
    from foodrec_amd import ScoringEngine
    U, I, C, B = 3000, 500, 4, 40000
    PM, RE, CE, users, items, _ = random_case(U, I, C, E, B, seed=E + 5)
    rng = np.random.default_rng(E)
    dish_cats = rng.integers(0, 2, (I, C)).astype(np.float32)
    dish_cats[dish_cats.sum(1) == 0, 1] = 1
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(dish_cats)
    eng.set_mlp_head(*_head((C + 1) * E, 256, 64, rng, scale=3.0))
    ut, it = torch.as_tensor(users, device="cuda"), torch.as_tensor(items, device="cuda")
    a = eng.score_pairs_mlp(ut, it); eng.check()
    assert eng.last_kernel() == "m2d_mlp_pc_bf16x3"
    eng.set_option("variant", 16)
    b = eng.score_pairs_mlp(ut, it); eng.check()
    assert eng.last_kernel() == "m2d_mlp_pc_bf16x3"
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    bad = users.copy(); bad[123] = U + 7
    with pytest.raises(IndexError, match="user id %d at position 123" % (U + 7)):
        eng.score_pairs_mlp(torch.as_tensor(bad, device="cuda"), it); eng.check()
    eng.set_option("variant", 0)


@pytest.mark.parametrize("coef", [0.99] + COEFS)
def test_mlp_reduces_to_reference_and_reports_bad_ids(coef):
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, C, E, B = 200, 100, 4, 128, 2000
    PM, RE, CE, users, items, _ = random_case(U, I, C, E, B, seed=1)
    dish_cats = np.random.default_rng(3).integers(0, 2, (I, C)).astype(np.float32)
    dish_cats[dish_cats.sum(1) == 0, 2] = 1
    rng = np.random.default_rng(2)
    W1, b1, W2, b2, w3, b3 = _head((C + 1) * E, 256, 64, rng)
    eng = ScoringEngine(PM, RE, CE, coef=coef); eng.set_dish_categories(dish_cats)
    eng.set_mlp_head(W1, b1, W2, b2, np.zeros_like(w3), 0.0)          # head contributes exactly 0
    ut, it = torch.as_tensor(users, device="cuda"), torch.as_tensor(items, device="cuda")
    got = eng.score_pairs_mlp(ut, it).cpu().numpy(); eng.check()
    assert_scores_close(got, oracle.inference_f64(PM, RE, CE, users, items, dish_cats[items], coef), 1e-5, "zero head")
    assert_scores_close(got, eng.score_pairs_bydish(ut, it).cpu().numpy(), 1e-5, "vs reference kernel")
    bad = users.copy(); bad[77] = U
    with pytest.raises(IndexError, match="user id %d at position 77" % U):
        eng.score_pairs_mlp(torch.as_tensor(bad, device="cuda"), it); eng.check()
    eng.clear_mlp_head()
    with pytest.raises(ValueError):
        eng.score_pairs_mlp(ut, it)


def test_mlp_with_ingredient_table():
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, C, E, R, B = 100, 80, 4, 64, 50, 1000
    PM, RE, CE, users, items, _ = random_case(U, I, C, E, B, seed=9)
    rng = np.random.default_rng(5)
    dish_cats = np.ones((I, C), np.float32)
    ING = (rng.standard_normal((R, E)) / 8).astype(np.float32)
    lens = rng.integers(1, 12, I); off = np.zeros(I + 1, np.int32); off[1:] = np.cumsum(lens)
    ids = rng.integers(0, R, off[-1]).astype(np.int32)
    head = _head((C + 1) * E, 256, 64, rng, scale=3.0)
    eng = ScoringEngine(PM, RE, CE); eng.set_dish_categories(dish_cats)
    eng.set_ingredients(ING, off, ids); eng.set_mlp_head(*head)
    got = eng.score_pairs_mlp(torch.as_tensor(users, device="cuda"), torch.as_tensor(items, device="cuda")); eng.check()
    H = oracle.dish_high_vectors(ING, off, ids)
    assert_scores_close(got.cpu().numpy(), oracle.inference_mlp(PM, RE, CE, dish_cats, *head, users, items, dish_high=H))


@pytest.mark.parametrize("E,B", [(128, 300_001), (64, 200_000), (32, 150_017)])
def test_mlp_many_tiles_per_block(E, B):
    """Batches of several tiles per workgroup: the producer / consumer kernel's steady state (row requests, ring stages
    and z sets that cross tile boundaries, the id conversion two tiles ahead) -- the small cases above run one tile per
    workgroup.  Every score against the every-wave-gathers kernel (an independent implementation), a sample against the
    float64 restatement, and an out-of-range id deep in the batch."""
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    C = 4 if E != 32 else 5
    U, I = 5000, 3000
    PM, RE, CE, users, items, _ = random_case(U, I, C, E, B, seed=E)
    rng = np.random.default_rng(E + 7)
    dish_cats = rng.integers(0, 2, (I, C)).astype(np.float32)
    dish_cats[dish_cats.sum(1) == 0, 1] = 1
    dish_cats[::5] *= rng.uniform(0.25, 3.0, (len(dish_cats[::5]), C)).astype(np.float32)     # weighted masks: the pattern is "weight != 0"
    dish_cats[11] = 0                                                                          # a dish without categories: NaN, its own bucket
    head = _head((C + 1) * E, 256, 64, rng, scale=4.0)
    coef = {128: 0.99, 64: 0.5, 32: 1.25}[E]
    eng = ScoringEngine(PM, RE, CE, coef=coef); eng.set_dish_categories(dish_cats); eng.set_mlp_head(*head)
    ut, it = torch.as_tensor(users, device="cuda"), torch.as_tensor(items, device="cuda")
    got = eng.score_pairs_mlp(ut, it); eng.check()
    assert eng.last_kernel() == "m2d_mlp_pc_bf16x3"
    # the same kernel without the grouping by mask pattern (every k-block of every tile): the skipped terms are zeros
    eng.set_option("skip_masked", 0)
    plain = eng.score_pairs_mlp(ut, it).cpu().numpy(); eng.check()
    eng.set_option("skip_masked", 1)
    gg = got.cpu().numpy()
    assert np.array_equal(np.isnan(gg), np.isnan(plain)) and np.isnan(gg).sum() == (items == 11).sum()
    assert np.nanmax(np.abs(gg - plain) / np.maximum(1.0, np.abs(plain))) < 2e-6
    eng.set_option("mlp_form", 1)
    other = eng.score_pairs_mlp(ut, it); eng.check()
    assert eng.last_kernel() == "m2d_mlp_mfma_bf16x3"
    eng.set_option("mlp_form", 0)
    g, o = got.cpu().numpy(), other.cpu().numpy()
    ok = items != 11
    assert np.isfinite(g[ok]).all() and np.isnan(o[~ok]).all()
    assert np.max(np.abs(g[ok] - o[ok]) / np.maximum(1.0, np.abs(o[ok]))) < 5e-5      # two split-bf16 kernels, different summation orders
    pick = np.concatenate([np.arange(0, 300), rng.integers(0, B, 3000), np.arange(B - 300, B)])
    ref = oracle.inference_mlp(PM, RE, CE, dish_cats, *head, users[pick], items[pick], coef=coef)
    assert_scores_close(g[pick], ref, what="E%d sample" % E)                # NaN rows (dish 11) must agree too
    bad = items.copy(); pos = B - 12_345; bad[pos] = I + 3
    with pytest.raises(IndexError, match="item id %d at position %d" % (I + 3, pos)):
        out = eng.score_pairs_mlp(ut, torch.as_tensor(bad, device="cuda")); eng.check()
