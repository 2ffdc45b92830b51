"""GPU parity of the batched evaluator (m2d_rank_candidates) with the reference's evaluate.py,
through fixtures that the reference's own code produced (tests/golden/make_reference_eval_golden.py)."""
import types

import numpy as np
import pytest

from helpers import load_json

pytestmark = pytest.mark.gpu

MODEL = load_json("ref_eval_model_cases.json")["cases"]


def _model(case):
    from foodrec_amd import Model
    PM, RE, CE = (np.asarray(case[k], dtype=np.float32) for k in ("PM", "RE", "CE"))
    args = types.SimpleNamespace(num_categories=4, num_users=PM.shape[0], embed_size=PM.shape[2],
                                 high_level_score_coefficient=0.99)
    return Model(args, PM, RE, CE, None), PM, RE, CE


@pytest.mark.parametrize("case", MODEL, ids=lambda c: "E%d-K%d" % (c["E"], c["K"]))
def test_evaluate_model_matches_reference(case):
    from foodrec_amd import Session, evaluate_model, eval_one_rating
    model, PM, RE, CE = _model(case)
    hits, ndcgs = evaluate_model(Session(model), model, case["testRatings"], case["testNegatives"], case["K"],
                                 case["dish_to_category"])
    assert hits == case["hits"]
    assert ndcgs == case["ndcgs"]
    # the reference-shaped one-call-per-user path gives the same answers
    for (u, hr, nd) in zip(case["testRatings"], case["hits"], case["ndcgs"]):
        assert eval_one_rating(model, u, case["testRatings"], case["testNegatives"], case["K"],
                               case["dish_to_category"]) == (hr, nd)


def test_rank_candidates_against_oracle_ranking():
    """Random scores with forced duplicates and ties, ragged lengths, several k."""
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    rng = np.random.default_rng(31)
    U, I, E = 40, 120, 32
    q = lambda shape: (rng.integers(-4, 5, shape) / 4.0).astype(np.float32)
    PM, RE, CE = q((U, 5, E)), q((I, E)), q((4, E))
    RE[60:80] = RE[0:20]
    pat = rng.integers(1, 16, I); pat[60:80] = pat[0:20]
    dish_cats = ((pat[:, None] >> np.arange(4)[None, :]) & 1).astype(np.float32)
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(dish_cats)
    for L, k in [(51, 10), (7, 10), (64, 1), (65, 64), (200, 33), (1024, 10)]:
        nseg = 37
        users = rng.integers(0, U, nseg).astype(np.int32)
        items = rng.integers(0, I, (nseg, L)).astype(np.int32)
        lens = rng.integers(1, L + 1, nseg).astype(np.int32)
        lens[0] = L
        s, ids, flags = eng.rank_candidates(torch.as_tensor(users, device="cuda"), torch.as_tensor(items, device="cuda"),
                                            k, lens=torch.as_tensor(lens, device="cuda"))
        eng.check()
        ids = ids.cpu().numpy(); s = s.cpu().numpy()
        assert not flags.cpu().numpy().any()
        for r in range(nseg):
            cand = items[r, :lens[r]].tolist()
            sc = oracle.inference_f32(PM, RE, CE, [users[r]] * len(cand), cand, dish_cats[cand])
            want = oracle.rank_candidates(cand, sc, k)
            got = [int(x) for x in ids[r] if x >= 0]
            assert got == [int(x) for x in want], (L, k, r)
            table = dict(zip(cand, sc))
            assert np.array_equal(s[r, :len(got)], np.array([table[i] for i in got], dtype=np.float32))
            assert np.all(ids[r, len(got):] == -1)


def test_nan_segments_are_flagged():
    import torch
    from foodrec_amd import ScoringEngine
    rng = np.random.default_rng(2)
    PM, RE, CE = (rng.standard_normal(s).astype(np.float32) for s in ((8, 5, 64), (30, 64), (4, 64)))
    dish_cats = np.ones((30, 4), np.float32); dish_cats[11] = 0
    eng = ScoringEngine(PM, RE, CE); eng.set_dish_categories(dish_cats)
    items = np.tile(np.arange(10, dtype=np.int32), (3, 1)); items[1, 4] = 11
    _, _, flags = eng.rank_candidates(torch.zeros(3, dtype=torch.int32, device="cuda"),
                                      torch.as_tensor(items, device="cuda"), 5)
    eng.check()
    assert flags.cpu().numpy().tolist() == [0, 1, 0]
