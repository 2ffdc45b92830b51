"""BASELINE configs[2], configs[3] and configs[4] at their full table sizes on ONE GPU (the 8-GPU aspect is the user-range shard:
an engine with a non-zero user_base holding its slice).  The oracle cannot score tables of this size, so the checks
are the size-independent ones: a random sample of pairs against the float64 restatement on the gathered rows,
permutation invariance, exact linearity under power-of-two scaling, and -- for retrieval -- agreement of the returned
lists with an exhaustive scoring of the same users by the exact pair kernel plus the oracle on the returned ids."""
import numpy as np
import pytest

from helpers import TOL, assert_scores_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    yield torch
    torch.cuda.empty_cache()                                 # 13 GB tables: hand them back before the next module


def _tables(torch, U, I, C, E, seed):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    s = 1.0 / np.sqrt(E)
    PM = torch.randn((U, C + 1, E), generator=g, device="cuda") * s
    RE = torch.randn((I, E), generator=g, device="cuda") * s
    CE = torch.randn((C, E), generator=g, device="cuda") * s
    pat = torch.randint(1, 2 ** C, (I,), generator=g, device="cuda", dtype=torch.int32)
    dish_cats = ((pat[:, None] >> torch.arange(C, device="cuda", dtype=torch.int32)[None, :]) & 1).float()
    return g, PM, RE, CE, dish_cats


def _check_retrieval(torch, eng, PM, RE, CE, dish_cats, users_local, base, k, exhaustive):
    """top-k of `users_local` (+ base) against (i) the oracle on the returned ids, (ii) for the first `exhaustive`
    users, every dish scored by the exact pair kernel: the k-th returned score is the k-th best within tolerance."""
    from oracle import m2d_oracle as oracle
    I = RE.shape[0]
    ut = (users_local + base).to(torch.int32)
    s, ids = eng.topk_users(ut, k); eng.check()
    assert int(ids.min()) >= 0 and int(ids.max()) < I
    s_h, ids_h, ul = s.cpu().numpy(), ids.cpu().numpy().astype(np.int64), users_local.cpu().numpy()
    n = len(ul)
    # (i) oracle on the returned (user, dish) pairs, rows gathered on the device
    flat_u = np.repeat(np.arange(n), k)
    ref = oracle.inference_f64(PM[users_local.long()].cpu().numpy(), RE[ids.reshape(-1).long()].cpu().numpy(), CE.cpu().numpy(),
                               flat_u, np.arange(n * k), dish_cats[ids.reshape(-1).long()].cpu().numpy())
    assert_scores_close(s_h.reshape(-1), ref, what="returned scores")
    assert np.all(s_h[:, :-1] >= s_h[:, 1:]), "lists not in descending order"
    assert all(len(set(row.tolist())) == k for row in ids_h), "duplicate dish in a list"
    # (ii) exhaustive: all I dishes of a few users through the exact-f32 pair kernel (resident masks)
    all_items = torch.arange(I, dtype=torch.int32, device="cuda")
    for r in range(exhaustive):
        full = eng.score_pairs_bydish(torch.full((I,), int(ut[r]), dtype=torch.int32, device="cuda"), all_items)
        best = torch.topk(full, k).values.cpu().numpy()
        assert np.all(np.abs(best - s_h[r]) <= TOL * np.maximum(1.0, np.abs(best))), (r, best, s_h[r])
        # every returned dish really has (about) the score reported for it
        assert np.all(np.abs(full[ids[r].long()].cpu().numpy() - s_h[r]) <= TOL * np.maximum(1.0, np.abs(s_h[r])))
    eng.check()


def test_config3_tables_10M_users_1M_dishes(torch_cuda):
    """configs[3]: 10 M users x 1 M dishes, E = 64 -- Personal_Memory 12.8 GB + Recipe_Embedding 256 MB on one GPU,
    held as the shard [30 M, 40 M) of a larger id space."""
    torch = torch_cuda
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, C, E, B, base = 10_000_000, 1_000_000, 4, 64, 1 << 22, 30_000_000
    g, PM, RE, CE, dish_cats = _tables(torch, U, I, C, E, 20260104)
    users_l = torch.randint(0, U, (B,), generator=g, device="cuda", dtype=torch.int32)
    items = torch.randint(0, I, (B,), generator=g, device="cuda", dtype=torch.int32)
    pat = torch.randint(1, 16, (B,), generator=g, device="cuda", dtype=torch.int32)
    cats = ((pat[:, None] >> torch.arange(4, device="cuda", dtype=torch.int32)[None, :]) & 1).float()
    eng = ScoringEngine(PM, RE, CE, user_base=base)
    users = users_l + base                                   # global ids
    out = eng.score_pairs(users, items, cats); eng.check()
    assert torch.isfinite(out).all()
    idx = torch.randint(0, B, (4096,), generator=g, device="cuda")
    su, si = users_l[idx].long(), items[idx].long()
    ref = oracle.inference_f64(PM[su].cpu().numpy(), RE[si].cpu().numpy(), CE.cpu().numpy(), np.arange(4096),
                               np.arange(4096), cats[idx].cpu().numpy())
    assert_scores_close(out[idx].cpu().numpy(), ref, what="config-3 sample")
    perm = torch.randperm(B, generator=g, device="cuda")
    out_p = eng.score_pairs(users[perm].contiguous(), items[perm].contiguous(), cats[perm].contiguous()); eng.check()
    assert torch.equal(out_p, out[perm])
    # an id of another shard is refused, not wrapped
    with pytest.raises(IndexError, match="user id %d" % (base - 1)):
        eng.score_pairs(torch.tensor([base - 1], dtype=torch.int32, device="cuda"), items[:1], cats[:1]); eng.check()
    # retrieval over the 1 M-dish catalogue for users of this shard (split-bf16 kernel), then the exact-f32 form
    eng.set_dish_categories(dish_cats)
    tk = torch.randperm(U, generator=g, device="cuda")[:512]
    _check_retrieval(torch, eng, PM, RE, CE, dish_cats, tk, base, 10, exhaustive=4)
    eng.set_option("topk_bf16x3", 0)
    _check_retrieval(torch, eng, PM, RE, CE, dish_cats, tk[:128], base, 10, exhaustive=2)
    eng.set_option("topk_bf16x3", 1)
    # linearity, in place (a second 12.8 GB table is not needed): score(2 PM) = 2 score(PM) exactly
    PM.mul_(2.0); eng.tables_updated()
    out2 = eng.score_pairs(users, items, cats); eng.check()
    assert torch.equal(out2, out * 2)


def test_config4_retrieval_e128_1M_dishes(torch_cuda):
    """configs[4]: full-catalogue top-10, E = 128, 1 M dishes (1 M users = one GPU's shard of the 8-way job), both
    kernel forms of the split-bf16 path and the exact-f32 path."""
    torch = torch_cuda
    from foodrec_amd import ScoringEngine
    U, I, C, E, base = 1_000_000, 1_000_000, 4, 128, 5_000_000
    g, PM, RE, CE, dish_cats = _tables(torch, U, I, C, E, 20260105)
    eng = ScoringEngine(PM, RE, CE, user_base=base)
    eng.set_dish_categories(dish_cats)
    tk = torch.randperm(U, generator=g, device="cuda")[:384]
    _check_retrieval(torch, eng, PM, RE, CE, dish_cats, tk, base, 10, exhaustive=4)
    for form in (1, 2):                                      # first form / pipelined form, forced
        eng.set_option("topk_form", form)
        _check_retrieval(torch, eng, PM, RE, CE, dish_cats, tk[:128], base, 10, exhaustive=1)
    eng.set_option("topk_form", 0)
    eng.set_option("topk_bf16x3", 0)
    _check_retrieval(torch, eng, PM, RE, CE, dish_cats, tk[:96], base, 16, exhaustive=1)
    # many users: one full round of blocks (65 536 users x 1 M dishes); lists of the first 64 users re-checked
    eng.set_option("topk_bf16x3", 1)
    big = torch.randperm(U, generator=g, device="cuda")[:65536]
    s, ids = eng.topk_users((big + base).to(torch.int32), 10); eng.check()
    s2, ids2 = eng.topk_users((big[:64] + base).to(torch.int32), 10); eng.check()
    assert torch.equal(ids[:64], ids2) and torch.equal(s[:64], s2)      # a user's list does not depend on the batch it is in


def test_config2_mlp_head_1M_users_e128(torch_cuda):
    """BASELINE configs[2] at its table sizes: 1 M users x 100 k dishes, E = 128, the 3-layer head on 2 M pairs.  Size-
    independent checks of the producer / consumer kernel with its pairs regrouped by mask pattern: a sample against the
    float64 restatement on gathered rows, permutation invariance (the grouping moves every pair), and a head that
    contributes nothing reducing to the exact pair kernel."""
    torch = torch_cuda
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, C, E, B = 1_000_000, 100_000, 4, 128, 1 << 21
    g, PM, RE, CE, dish_cats = _tables(torch, U, I, C, E, seed=31)
    K = (C + 1) * E
    rn = lambda *shape: torch.randn(shape, generator=g, device="cuda")
    head = (rn(K, 256) * (4 / K ** 0.5), rn(256) * 0.1, rn(256, 64) / 4, rn(64) * 0.1, rn(64) / 2, 0.125)
    eng = ScoringEngine(PM, RE, CE); eng.set_dish_categories(dish_cats); eng.set_mlp_head(*head)
    users = torch.randint(0, U, (B,), generator=g, device="cuda", dtype=torch.int32)
    items = torch.randint(0, I, (B,), generator=g, device="cuda", dtype=torch.int32)
    got = eng.score_pairs_mlp(users, items); eng.check()
    assert eng.last_kernel() == "m2d_mlp_pc_bf16x3" and bool(torch.isfinite(got).all())
    # (i) sample against the restatement, rows gathered on the device
    pick = torch.randint(0, B, (4096,), generator=g, device="cuda")
    pu, pd = users[pick].long(), items[pick].long()
    hd = [x.cpu().numpy() if hasattr(x, "cpu") else x for x in head]
    ref = oracle.inference_mlp(PM[pu].cpu().numpy(), RE[pd].cpu().numpy(), CE.cpu().numpy(), dish_cats[pd].cpu().numpy(), *hd,
                               np.arange(4096), np.arange(4096))
    assert_scores_close(got[pick].cpu().numpy(), ref, what="sample")
    # (ii) permutation invariance
    perm = torch.randperm(B, generator=g, device="cuda")
    again = eng.score_pairs_mlp(users[perm].contiguous(), items[perm].contiguous()); eng.check()
    d = (again - got[perm]).abs() / got[perm].abs().clamp(min=1.0)
    assert float(d.max()) < 2e-6
    # (iii) a head that contributes nothing: the reference score, as the exact pair kernel computes it
    eng.set_mlp_head(head[0], head[1], head[2], head[3], torch.zeros_like(head[4]), 0.0)
    base = eng.score_pairs_mlp(users[: 1 << 19], items[: 1 << 19]); eng.check()
    exact = eng.score_pairs_bydish(users[: 1 << 19], items[: 1 << 19]); eng.check()
    assert float(((base - exact).abs() / exact.abs().clamp(min=1.0)).max()) < 1e-5
