#!/usr/bin/env python3
"""One-off randomized soak of pair scoring, candidate ranking, catalogue retrieval, Write_Memory, the training step and the
ingredient table against the
restatements (test infrastructure, not collected by pytest: seeds come from the clock).
Usage on the GPU box: python tests/soak_random_cases.py [cases]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from foodrec_amd import ScoringEngine
from oracle import m2d_oracle as oracle
from helpers import TOL, assert_scores_close

dev = lambda a: torch.as_tensor(a, device="cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())      # [cases] [first seed]
print("seed0", seed0)


def tables(rng, U, I, C, E):
    s = 1.0 / np.sqrt(E)
    return ((rng.standard_normal((U, C + 1, E)) * s).astype(np.float32), (rng.standard_normal((I, E)) * s).astype(np.float32),
            (rng.standard_normal((C, E)) * s).astype(np.float32))


def masks(rng, I, C, weighted, empty):
    m = rng.integers(0, 2, (I, C)).astype(np.float32)
    if not empty:
        m[m.sum(1) == 0, rng.integers(0, C)] = 1
    if weighted:
        m *= rng.uniform(0.1, 3.0, (I, C)).astype(np.float32)
    return m


def pairs_case(rng, what):
    C = int(rng.choice([4, 4, 4, 3, 6])); E = int(rng.choice([32, 64, 128, 200, 24, 256, 8]))
    U = int(rng.integers(1, 3000)); I = int(rng.integers(1, 500))
    B = int(rng.choice([1, 51, int(rng.integers(2, 9000)), int(rng.integers(9000, 400000))]))
    PM, RE, CE = tables(rng, U, I, C, E)
    dc = masks(rng, I, C, rng.integers(0, 2), rng.integers(0, 2))
    users = rng.integers(0, U, B).astype(np.int32); items = rng.integers(0, I, B).astype(np.int32)
    eng = ScoringEngine(PM, RE, CE, coef=COEF); eng.set_dish_categories(dc)
    opts = {"skip_masked": int(rng.integers(0, 2)), "prefetch": int(rng.choice([1, 2, 4])), "nt_loads": int(rng.integers(0, 2)),
            "user_high_table": int(rng.integers(0, 2))}
    for k, v in opts.items():
        eng.set_option(k, v)
    by_dish = bool(rng.integers(0, 2))
    got = (eng.score_pairs_bydish(dev(users), dev(items)) if by_dish else eng.score_pairs(dev(users), dev(items), dev(dc[items])))
    eng.check()
    pick = rng.integers(0, B, min(B, 4000))
    ref = oracle.inference_f64(PM, RE, CE, users[pick], items[pick], dc[items[pick]], COEF)
    assert_scores_close(got.cpu().numpy()[pick], ref, what="%s C%d E%d B%d coef %s %s" % (what, C, E, B, COEF, opts))
    host = eng.score_pairs_host(users[:70000], items[:70000], dc[items[:70000]])
    if not opts["user_high_table"] or B < 2 ** 18:
        assert np.array_equal(host, got.cpu().numpy()[:70000], equal_nan=True), what
    return "pairs C%d E%d B%d %s" % (C, E, B, eng.last_kernel())


def rank_case(rng, what):
    C, E = 4, int(rng.choice([32, 64, 200, 128]))
    U = int(rng.integers(1, 500)); I = int(rng.integers(2, 400))
    nseg = int(rng.integers(1, 700)); L = int(rng.choice([1, 51, 100, int(rng.integers(2, 1025))])); k = int(rng.integers(1, 65))
    PM, RE, CE = tables(rng, U, I, C, E)
    dc = masks(rng, I, C, rng.integers(0, 2), rng.integers(0, 2))
    users = rng.integers(0, U, nseg).astype(np.int32); items = rng.integers(0, I, (nseg, L)).astype(np.int32)
    lens = rng.integers(1, L + 1, nseg).astype(np.int32) if rng.integers(0, 2) else None
    eng = ScoringEngine(PM, RE, CE, coef=COEF); eng.set_dish_categories(dc)
    s, ids, flags = eng.rank_candidates(dev(users), dev(items), k, lens=dev(lens) if lens is not None else None); eng.check()
    s, ids = s.cpu().numpy(), ids.cpu().numpy()
    for r in rng.integers(0, nseg, min(nseg, 40)):
        n_r = int(lens[r]) if lens is not None else L
        cand = items[r, :n_r]
        sc = eng.score_pairs_bydish(dev(np.full(n_r, users[r], np.int32)), dev(cand)).cpu().numpy()   # same kernel arithmetic
        if np.isnan(sc).any():
            continue                                            # NaN ordering under heapq is input-order dependent: covered by the fixtures
        want = oracle.rank_candidates(cand.tolist(), sc.tolist(), k)
        got = [int(x) for x in ids[r] if x >= 0]
        assert got == [int(x) for x in want], (what, r, got, want)
    return "rank E%d nseg%d L%d k%d" % (E, nseg, L, k)


def topk_case(rng, what):
    C = 4; E = int(rng.choice([32, 64, 128, 200, 16]))
    U = int(rng.integers(1, 400)); I = int(rng.integers(1, 6000)); k = int(rng.integers(1, min(64, I) + 1))     # the call needs k <= I
    PM, RE, CE = tables(rng, U, I, C, E)
    dc = masks(rng, I, C, rng.integers(0, 4) == 0, rng.integers(0, 2))
    eng = ScoringEngine(PM, RE, CE, coef=COEF); eng.set_dish_categories(dc)
    for name in ("topk_bf16x3", "topk_grouped"):
        eng.set_option(name, int(rng.integers(0, 4) != 0))
    eng.set_option("topk_form", int(rng.integers(0, 5)))
    nU = int(rng.integers(1, 300))
    users = rng.integers(0, U, nU).astype(np.int32)
    s, ids = eng.topk_users(dev(users), k); eng.check()
    s, ids = s.cpu().numpy(), ids.cpu().numpy()
    all_items = np.arange(I); nv = min(k, I)
    for r in rng.integers(0, nU, min(nU, 12)):
        ref = oracle.inference_f64(PM, RE, CE, np.full(I, users[r]), all_items, dc, COEF)
        g = ids[r, :nv]
        assert np.all(g >= 0) and np.all(g < I) and len(set(g.tolist())) == nv, (what, g)
        assert_scores_close(s[r, :nv], ref[g], what=what)
        key = np.where(np.isnan(s[r, :nv]), -np.inf, s[r, :nv])
        assert np.all(key[:-1] >= key[1:]), what
        rest = np.delete(np.where(np.isnan(ref), -np.inf, ref), g)
        if rest.size and np.isfinite(key[nv - 1]):
            assert rest.max() <= key[nv - 1] + TOL * max(1.0, abs(key[nv - 1])), (what, rest.max(), key[nv - 1])
        for a in range(nv - 1):
            if s[r, a] == s[r, a + 1]:
                assert g[a] < g[a + 1], what
    return "topk E%d I%d k%d nU%d %s" % (E, I, k, nU, eng.last_kernel())


def write_case(rng, what):
    C = int(rng.choice([4, 3])); E = int(rng.choice([32, 64, 200, 6]))
    U = int(rng.integers(1, 300)); I = int(rng.integers(1, 100)); L = int(rng.choice([95, 7, 130, 300, 1]))
    B = int(rng.choice([256, int(rng.integers(1, 2049)), int(rng.integers(2049, 6000))]))
    PM, RE, CE = tables(rng, U, I, C, E)
    dc = masks(rng, I, C, rng.integers(0, 2), False)
    users = rng.integers(0, U, B).astype(np.int32); items = rng.integers(0, I, B).astype(np.int32)
    cats = dc[items]
    GM = (rng.standard_normal((L, C + 1, E)) / 4).astype(np.float32)
    sign = np.where(rng.random(B) < 0.6, 1.0, -1.0).astype(np.float32)
    y = (rng.random((B, L)) < 0.1).astype(np.float32)
    y[y.sum(1) == 0, 0] = 1
    eng = ScoringEngine(PM, RE, CE)
    gm = dev(GM).clone()
    eng.write_memory(dev(users), dev(items), dev(cats), dev(sign), dev(y), gm, 0.01, 0.02, 0.03); eng.check()
    PM2, GM2, _, _ = oracle.write_memory(PM, RE, CE, GM, users, items, cats, sign, y, 0.01, 0.02, 0.03)
    for got, ref in ((eng.pm.cpu().numpy(), PM2), (gm.cpu().numpy(), GM2)):
        got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), what
        ok = ~np.isnan(ref)
        err = np.abs(got[ok] - ref[ok]) / np.maximum(1.0, np.abs(ref[ok]))
        assert err.size == 0 or err.max() <= 5e-5, (what, err.max())
    return "write C%d E%d L%d B%d" % (C, E, L, B)


def train_case(rng, what):
    from oracle import train_oracle as T
    C = int(rng.choice([4, 3])); E = int(rng.choice([32, 64, 200, 6, 128]))
    U = int(rng.integers(2, 3000)); I = int(rng.integers(2, 600)); B = int(rng.choice([256, int(rng.integers(1, 300)), int(rng.integers(300, 9000))]))
    learner = str(rng.choice(["adam", "sgd", "adagrad", "rmsprop"])); lr = 0.01
    PM, RE, CE = tables(rng, U, I, C, E)
    PM, RE, CE = PM * 3, RE * 3, CE * 3
    eng = ScoringEngine(PM.copy(), RE.copy(), CE.copy()); eng.train_begin(learner, lr)
    st = T.TrainState(PM, RE, CE, learner, lr)
    steps = 2
    for _ in range(steps):
        users = rng.integers(0, U, B).astype(np.int32); items = rng.integers(0, I, B).astype(np.int32)
        users[: B // 4] = users[0]
        cats = rng.integers(0, 2, (B, C)).astype(np.float32)
        cats[cats.sum(1) == 0, rng.integers(0, C)] = 1.0
        if rng.integers(0, 2): cats *= rng.uniform(0.5, 2.0, (B, C)).astype(np.float32)
        labels = rng.integers(0, 2, B).astype(np.float32)
        ref_loss, ref_norm = st.step(users, items, cats, labels)
        loss, norm, _, _ = eng.train_step(dev(users), dev(items), dev(cats), dev(labels)).cpu().numpy(); eng.check()
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (what, loss, ref_loss)
        assert abs(norm - ref_norm) <= 2e-5 * max(1.0, ref_norm), (what, norm, ref_norm)
    # adam / rmsprop divide by sqrt(v) + 1e-8: an element whose gradient all but cancels in float32 moves by a different fraction of lr than
    # in the float64 restatement.  The general bound stays tests/test_gpu_train.py's 1e-3 of lr per step; a cancelling element may reach 1.5e-3
    # (seed 1791184991 case 124: ONE element at 1.0003e-3) -- but only a handful may: an optimiser regression moves whole rows, and a bound
    # widened for every element would let it through (round 5's form of this check)
    adaptive = learner in ("adam", "rmsprop")
    tol = 1e-3 * lr * steps if adaptive else 1e-5
    for got, ref in ((eng.pm, st.PM), (eng.re, st.RE), (eng.ce, st.CE)):
        err = np.abs(got.cpu().numpy().astype(np.float64) - ref)
        bound = tol * np.maximum(1.0, np.abs(ref)) if learner in ("sgd", "adagrad") else tol
        if adaptive:
            over = err > bound
            assert np.all(err <= 1.5 * bound) and over.sum() <= max(2, err.size // 100000), (what, learner, err.max(), int(over.sum()))
        else:
            assert np.all(err <= bound), (what, learner, err.max())
    eng.train_end()
    return "train %s C%d E%d B%d" % (learner, C, E, B)


def ingredients_case(rng, what):
    C = 4; E = int(rng.choice([6, 32, 64, 128, 200, 320]))
    U = int(rng.integers(1, 500)); I = int(rng.integers(1, 400)); R = int(rng.integers(1, 600)); B = int(rng.integers(1, 20000))
    PM, RE, CE = tables(rng, U, I, C, E)
    ING = (rng.standard_normal((R, E)) / np.sqrt(E)).astype(np.float32)
    lens = rng.integers(0 if rng.integers(0, 2) else 1, int(rng.integers(1, 40)) + 1, I)
    off = np.zeros(I + 1, np.int32); off[1:] = np.cumsum(lens)
    ids = rng.integers(0, R, off[-1]).astype(np.int32)
    w = rng.uniform(0.5, 2.0, len(ids)).astype(np.float32) if rng.integers(0, 2) else None
    dc = masks(rng, I, C, rng.integers(0, 2), rng.integers(0, 2))
    users = rng.integers(0, U, B).astype(np.int32); items = rng.integers(0, I, B).astype(np.int32)
    eng = ScoringEngine(PM, RE, CE); eng.set_ingredients(ING, off, ids, w); eng.set_dish_categories(dc)
    got = eng.score_pairs_ingredients(dev(users), dev(items)); eng.check()
    pick = rng.integers(0, B, min(B, 3000))
    ref = oracle.inference_ingredients(PM, RE, ING, off, ids, w, users[pick], items[pick], dc[items[pick]])
    assert_scores_close(got.cpu().numpy()[pick], ref, what=what)
    return "ingredients E%d I%d R%d B%d %s" % (E, I, R, B, eng.last_kernel())


kinds = [pairs_case, rank_case, topk_case, write_case, train_case, ingredients_case]
for it in range(n):
    rng = np.random.default_rng(seed0 + it)
    fn = kinds[it % len(kinds)]
    # --high_level_score_coefficient of the case (pairs / rank / topk; no draw: the seeds' tables stay what they were)
    COEF = [0.99, 0.5, 1.25, 0.0, 1.0, 0.9, 0.99][(it // len(kinds) + seed0) % 7]
    msg = fn(rng, "case %d seed %d" % (it, seed0 + it))
    if it % 12 < 6:
        print("ok", it, msg, flush=True)
print("all", n, "cases agree")
