"""CPU tests of the oracle itself: KAT, frozen golden vectors, cross-checks between the four
restatements (float64, float32-naive, factored, C), and properties of the formula."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from helpers import COEFS, assert_scores_close, random_case, score_cases
from oracle import c_oracle, m2d_oracle as oracle


def test_hand_known_answer():
    # SURVEY.md section 8a, derived by hand from Model_Recommender.py:67-96
    PM = np.array([[[1, 2], [1, 0], [0, 1], [2, 2], [3, -1]]], dtype=np.float32)
    RE = np.array([[0.5, -1]], dtype=np.float32)
    CE = np.array([[1, 1], [2, 0], [0, 2], [-1, 1]], dtype=np.float32)
    m = [[[1.0], [0.0], [1.0], [0.0]]]
    # high = <[1,2], [1,1]+[0,2]>/2 = 3.5 ; low = <[.5,-1], [1,0]+[2,2]>/2 = -0.25
    for fn in (oracle.inference_f64, oracle.inference_f32):
        assert abs(float(fn(PM, RE, CE, [0], [0], m)[0]) - 3.4625) < 1e-6
    assert abs(float(c_oracle.score_pairs(PM, RE, CE, [0], [0], [[1, 0, 1, 0]])[0]) - 3.4625) < 1e-6
    a, b = oracle.blend_coefficients(0.99)
    assert a.dtype == np.float32 and float(b) == pytest.approx(0.00999999046, abs=1e-11)


@pytest.mark.parametrize("path", score_cases(), ids=lambda p: p.split("score_")[-1][:-4])
def test_golden_vectors_frozen(path):
    z = np.load(path)
    args = (z["PM"], z["RE"], z["CE"], z["users"], z["items"], z["cats"])
    coef = float(z["coef"])                                # 0.99 (the flag's default) or the *_coef* cases' own
    f64 = oracle.inference_f64(*args, coef)
    f32 = oracle.inference_f32(*args, coef)
    assert_scores_close(f64, z["score_f64"], 1e-12, "f64 vs frozen")
    assert_scores_close(f32, z["score_f32"], 2e-6, "f32 vs frozen")        # numpy's pairwise sum may regroup
    assert_scores_close(f32, f64, 1e-5, "f32 vs f64")
    assert_scores_close(c_oracle.score_pairs(*args, coef=coef), f64, 1e-5, "C vs f64")
    assert_scores_close(c_oracle.score_pairs(*args, coef=coef, materialised=True), f64, 1e-5, "C-materialised vs f64")
    if "hand" in z.files:
        assert abs(float(f64[0]) - float(z["hand"][0])) < 1e-6


@pytest.mark.parametrize("shape", [(11, 13, 4, 6, 50), (257, 129, 4, 32, 400), (1000, 500, 4, 64, 2000),
                                   (1000, 500, 4, 128, 1000), (300, 100, 4, 200, 500), (20, 10, 7, 12, 64)])
def test_factored_form_agrees(shape):
    U, I, C, E, B = shape
    PM, RE, CE, users, items, _ = random_case(U, I, C, E, B, seed=sum(shape))
    rng = np.random.default_rng(5)
    dish_cats = rng.integers(0, 2, (I, C)).astype(np.float32)
    direct = oracle.inference_f64(PM, RE, CE, users, items, dish_cats[items])
    fact = oracle.inference_factored(PM, RE, CE, users, items, dish_cats)
    assert_scores_close(fact, direct, 1e-5, "factored vs direct")


@pytest.mark.parametrize("coef", COEFS)
def test_blend_coefficient_in_every_restatement(coef):
    """`coef * high + (1 - coef) * low` with a float32 `1 - coef` (Model_Recommender.py:17, :95-96): the float64, float32,
    factored, C and torch restatements agree at every coefficient, and the two levels can be read off the extremes."""
    import torch
    from oracle import torch_graph
    PM, RE, CE, users, items, _ = random_case(60, 40, 4, 32, 300, seed=int(coef * 100) + 3)
    dish_cats = np.random.default_rng(2).integers(0, 2, (40, 4)).astype(np.float32)
    cats = dish_cats[items]
    a, b = oracle.blend_coefficients(coef)
    assert float(b) == float(np.float32(1.0) - np.float32(coef))
    f64 = oracle.inference_f64(PM, RE, CE, users, items, cats, coef)
    assert_scores_close(oracle.inference_f32(PM, RE, CE, users, items, cats, coef), f64, 1e-5, "f32")
    assert_scores_close(oracle.inference_factored(PM, RE, CE, users, items, dish_cats, coef), f64, 1e-5, "factored")
    assert_scores_close(c_oracle.score_pairs(PM, RE, CE, users, items, cats, coef=coef), f64, 1e-5, "C")
    got = torch_graph.inference(*(torch.from_numpy(x) for x in (PM, RE, CE, users, items, cats)), coef=coef).numpy()
    assert_scores_close(got, f64, 1e-5, "torch graph")
    high = oracle.inference_f64(PM, RE, CE, users, items, cats, 1.0)       # 1 - 1 = 0: the high level alone
    low = oracle.inference_f64(PM, RE, CE, users, items, cats, 0.0)
    assert_scores_close(f64, float(a) * high + float(b) * low, 1e-12, "levels")


def test_zero_mask_is_nan_and_weights_are_linear():
    PM, RE, CE, users, items, cats = random_case(30, 20, 4, 16, 64, seed=3)
    cats[7] = 0
    s = oracle.inference_f32(PM, RE, CE, users, items, cats)
    assert np.isnan(s[7]) and np.isnan(c_oracle.score_pairs(PM, RE, CE, users, items, cats)[7])
    # scaling every weight of a row by the same factor leaves the score unchanged (sum/n)
    ok = cats.sum(1) > 0
    s2 = oracle.inference_f64(PM, RE, CE, users, items, cats * 2.5)
    assert_scores_close(s2[ok], oracle.inference_f64(PM, RE, CE, users, items, cats)[ok], 1e-12)


def test_linearity_in_personal_memory_and_batch_permutation():
    PM, RE, CE, users, items, cats = random_case(40, 30, 4, 24, 128, seed=9, zero_rows=False)
    PM2 = np.random.default_rng(1).standard_normal(PM.shape).astype(np.float32)
    s1 = oracle.inference_f64(PM, RE, CE, users, items, cats)
    s2 = oracle.inference_f64(PM2, RE, CE, users, items, cats)
    s12 = oracle.inference_f64(PM.astype(np.float64) * 2 + PM2.astype(np.float64) * -3, RE, CE, users, items, cats)
    assert_scores_close(s12, 2 * s1 - 3 * s2, 1e-9)
    perm = np.random.default_rng(2).permutation(len(users))
    assert np.array_equal(oracle.inference_f32(PM, RE, CE, users[perm], items[perm], cats[perm]),
                          oracle.inference_f32(PM, RE, CE, users, items, cats)[perm])


def test_feed_conversions_and_errors():
    PM, RE, CE, users, items, cats = random_case(12, 9, 4, 8, 10, seed=4, zero_rows=False)
    base = oracle.inference_f32(PM, RE, CE, users, items, cats)
    as_str = oracle.inference_f32(PM, RE, CE, [str(u) for u in users], [str(i) for i in items], cats[:, :, None].tolist())
    assert np.array_equal(base, as_str)
    with pytest.raises(IndexError):
        oracle.inference_f32(PM, RE, CE, [12], [0], cats[:1])
    with pytest.raises(IndexError):
        oracle.inference_f32(PM, RE, CE, [0], [-1], cats[:1])
    with pytest.raises(IndexError):
        c_oracle.score_pairs(PM, RE, CE, [0], [9], cats[:1])
    assert oracle.inference_f32(PM, RE, CE, [], [], np.zeros((0, 4))).shape == (0,)


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 6), st.integers(1, 40), st.integers(1, 30), st.integers(0, 2 ** 31 - 1))
def test_restatements_agree_property(C, E, B, seed):
    PM, RE, CE, users, items, cats = random_case(7, 5, C, E, B, seed, zero_rows=False)
    f64 = oracle.inference_f64(PM, RE, CE, users, items, cats)
    assert_scores_close(oracle.inference_f32(PM, RE, CE, users, items, cats), f64, 1e-5)
    assert_scores_close(c_oracle.score_pairs(PM, RE, CE, users, items, cats), f64, 1e-5)


def test_torch_graph_restatement_agrees():
    import torch
    from oracle import torch_graph
    PM, RE, CE, users, items, cats = random_case(200, 100, 4, 64, 1000, seed=14)
    got = torch_graph.inference(torch.from_numpy(PM), torch.from_numpy(RE), torch.from_numpy(CE),
                                torch.from_numpy(users), torch.from_numpy(items), torch.from_numpy(cats)).numpy()
    assert_scores_close(got, oracle.inference_f64(PM, RE, CE, users, items, cats), 1e-5)


def test_ingredient_extension_reduces_to_reference_in_the_oracle():
    """Build-defined superset: with ING = CE, ids = 0..C-1 and w = the mask it IS the
    reference formula."""
    PM, RE, CE, users, items, _ = random_case(40, 30, 4, 16, 200, seed=77)
    dish_cats = np.random.default_rng(1).integers(0, 2, (30, 4)).astype(np.float32)
    off = np.arange(31) * 4
    ids = np.tile(np.arange(4), 30)
    ext = oracle.inference_ingredients(PM, RE, CE, off, ids, dish_cats.reshape(-1), users, items, dish_cats[items])
    assert_scores_close(ext, oracle.inference_f64(PM, RE, CE, users, items, dish_cats[items]), 1e-12)


def test_write_memory_restatement_by_hand():
    """One pair, C = 2, E = 1: every term of Model_Recommender.py:106-215 can be followed by hand."""
    PM = np.zeros((2, 3, 1)); RE = np.array([[2.0]]); CE = np.array([[1.0], [3.0]])
    GM = np.array([[[10.0], [20.0], [30.0]], [[1.0], [2.0], [3.0]]])
    PM2, GM2, mp, mg = oracle.write_memory(PM, RE, CE, GM, [1], [0], [[1.0, 1.0]], [1.0], [[1.0, 1.0]],
                                           beta_1=0.5, beta_2=0.25, alpha=0.1)
    # v = [0.25*(1+3)/2, 0.5*1*2, 0.5*1*2] = [0.5, 1, 1]; g = mean over both labels = [5.5, 11, 16.5]
    assert np.allclose(PM2[1, :, 0], [0.5 + 0.55, 1 + 1.1, 1 + 1.65]) and np.all(PM2[0] == 0)
    assert np.allclose(GM2[:, :, 0], GM[:, :, 0] + [0.5, 1, 1])
    assert mp == pytest.approx(PM2.mean()) and mg == pytest.approx(GM2.mean())
