"""GPU parity of m2d_write_memory (scatter-add form of Model.Write_Memory, Model_Recommender.py:106-220)
with the op-for-op restatement that keeps the reference's dense one-hot matmuls."""
import numpy as np
import pytest

from helpers import random_case

pytestmark = pytest.mark.gpu


def _close(got, ref, tol=2e-5):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.all(np.abs(got[ok] - ref[ok]) <= tol * np.maximum(1.0, np.abs(ref[ok]))), np.abs(got[ok] - ref[ok]).max()


@pytest.mark.parametrize("U,I,C,E,L,B", [(50, 30, 4, 64, 95, 128), (20, 10, 4, 200, 7, 8), (9, 5, 3, 6, 4, 33), (300, 100, 4, 32, 95, 1000),
                                         (40, 30, 4, 64, 95, 3000),       # > 2048 pairs: the General_Memory assign by atomics
                                         (20, 10, 4, 32, 130, 2500),      # three label masks
                                         (20, 10, 4, 32, 300, 64)])       # more labels than the masks hold: every label walked
def test_write_memory_matches_restatement(U, I, C, E, L, B):
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    PM, RE, CE, users, items, cats = random_case(U, I, C, E, B, seed=U + B, zero_rows=False)
    rng = np.random.default_rng(L)
    GM = (rng.standard_normal((L, C + 1, E)) / 4).astype(np.float32)
    sign = np.where(rng.random(B) < 0.6, 1.0, -1.0).astype(np.float32)
    y = (rng.random((B, L)) < 0.1).astype(np.float32)
    y[y.sum(1) == 0, 0] = 1
    users[:4] = users[0]                                   # duplicate users accumulate (reduce_sum over the batch)
    eng = ScoringEngine(PM, RE, CE)
    gm = torch.as_tensor(GM, device="cuda").clone()
    means = eng.write_memory(torch.as_tensor(users, device="cuda"), torch.as_tensor(items, device="cuda"),
                             torch.as_tensor(cats, device="cuda"), torch.as_tensor(sign, device="cuda"),
                             torch.as_tensor(y, device="cuda"), gm, 0.01, 0.02, 0.03, want_means=True)
    PM2, GM2, mp, mg = oracle.write_memory(PM, RE, CE, GM, users, items, cats, sign, y, 0.01, 0.02, 0.03)
    _close(eng.pm.cpu().numpy(), PM2)
    _close(gm.cpu().numpy(), GM2)
    assert abs(means[0] - mp) < 1e-6 and abs(means[1] - mg) < 1e-6
    # the forward now scores with the written memory
    out = eng.score_pairs(torch.as_tensor(users, device="cuda"), torch.as_tensor(items, device="cuda"),
                          torch.as_tensor(cats, device="cuda")); eng.check()
    _close(out.cpu().numpy(), oracle.inference_f64(PM2, RE, CE, users, items, cats), 1e-4)


def test_write_memory_edge_cases():
    """0/0 cases.  The reference's DENSE one-hot matmuls multiply a NaN row by the zeros of every other
    user / label, so one pair with an empty mask turns the whole Personal_Memory (row 0) into NaN there.
    The scatter form touches only the rows the pair addresses: NaN lands in that user's block and in that
    pair's labels, nowhere else.  That containment is a deliberate, documented difference."""
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, C, E, L, B = 12, 8, 4, 64, 5, 6
    PM, RE, CE, _, items, cats = random_case(U, I, C, E, B, seed=3, zero_rows=False)
    users = np.arange(B, dtype=np.int32)                   # distinct users
    GM = np.random.default_rng(1).standard_normal((L, C + 1, E)).astype(np.float32)
    y = np.eye(L, dtype=np.float32)[np.arange(B) % L]
    sign = np.ones(B, np.float32)
    cats[2] = 0                                            # empty mask: v row 0 = 0/0
    y[3] = 0                                               # user with no label: g = 0/0
    eng = ScoringEngine(PM, RE, CE)
    gm = torch.as_tensor(GM, device="cuda").clone()
    t = lambda a: torch.as_tensor(a, device="cuda")
    eng.write_memory(t(users), t(items), t(cats), t(sign), t(y), gm, 0.01, 0.01, 0.01); eng.check()
    pm2, gm2 = eng.pm.cpu().numpy(), gm.cpu().numpy()
    assert np.isnan(pm2[2, 0]).all() and not np.isnan(pm2[2, 1:]).any()        # only row 0 of user 2
    assert np.isnan(pm2[3]).all()                                              # user 3: label mean is 0/0
    clean = [u for u in range(U) if u not in (2, 3)]
    assert not np.isnan(pm2[clean]).any()
    assert np.isnan(gm2[2 % L, 0]).all() and not np.isnan(gm2[[l for l in range(L) if l != 2 % L]]).any()
    # on the pairs without 0/0 the numbers are the reference's
    keep = np.array([0, 1, 4, 5])
    PMr, GMr, _, _ = oracle.write_memory(PM, RE, CE, GM, users[keep], items[keep], cats[keep], sign[keep], y[keep])
    _close(pm2[[0, 1, 4, 5]], PMr[[0, 1, 4, 5]])
    bad = users.copy(); bad[1] = U
    with pytest.raises(IndexError):
        eng.write_memory(t(bad), t(items), t(cats), t(sign), t(y), gm, 0.01, 0.01, 0.01); eng.check()
    eng.write_memory(t(users[:0]), t(items[:0]), t(cats[:0]), t(sign[:0]), t(y[:0]), gm, 0.01, 0.01, 0.01); eng.check()


@pytest.mark.parametrize("personal,general", [(True, False), (False, True)])
def test_write_memory_runs_only_the_fetched_assigns(personal, general):
    """`personal` depends on the two Personal_Memory assigns only (Model_Recommender.py:167, :198), `general` on the
    General_Memory assign only (:215); the driver's ordinary batch fetches `general` alone
    (Train_recommender.py:195-199) and must leave Personal_Memory bit for bit as it was."""
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, C, E, L, B = 40, 20, 4, 64, 9, 50
    PM, RE, CE, users, items, cats = random_case(U, I, C, E, B, seed=77, zero_rows=False)
    rng = np.random.default_rng(5)
    GM = (rng.standard_normal((L, C + 1, E)) / 4).astype(np.float32)
    sign = np.where(rng.random(B) < 0.5, 1.0, -1.0).astype(np.float32)
    y = (rng.random((B, L)) < 0.3).astype(np.float32); y[:, 1] = 1
    eng = ScoringEngine(PM, RE, CE)
    gm = torch.as_tensor(GM, device="cuda").clone()
    t = lambda a: torch.as_tensor(a, device="cuda")
    means = eng.write_memory(t(users), t(items), t(cats), t(sign), t(y), gm, 0.01, 0.02, 0.03, want_means=True,
                             write_pm=personal, write_gm=general)
    PM2, GM2, mp, mg = oracle.write_memory(PM, RE, CE, GM, users, items, cats, sign, y, 0.01, 0.02, 0.03,
                                           personal=personal, general=general)
    if personal:
        _close(eng.pm.cpu().numpy(), PM2)
        assert np.array_equal(gm.cpu().numpy(), GM) and means[1] is None and abs(means[0] - mp) < 1e-6
    else:
        _close(gm.cpu().numpy(), GM2)
        assert np.array_equal(eng.pm.cpu().numpy(), PM) and means[0] is None and abs(means[1] - mg) < 1e-6
    # a general-only call still refuses a bad id (the gathers are shared by both branches)
    bad = items.copy(); bad[3] = I
    with pytest.raises(IndexError):
        eng.write_memory(t(users), t(bad), t(cats), t(sign), t(y), gm, 0.01, 0.02, 0.03, write_pm=personal, write_gm=general)
        eng.check()
    with pytest.raises(ValueError):
        eng.write_memory(t(users), t(items), t(cats), t(sign), t(y), gm, 0.01, 0.02, 0.03, write_pm=False, write_gm=False)
