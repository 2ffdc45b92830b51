#!/usr/bin/env python3
"""Randomized soak of the retrieval call's plan -- scan-start bounds, pattern pruning, users sorted by mask, (user block,
dish range) items handed out longest first, thresholds shared between a user's dish ranges -- on calls large enough to use all
of it; every fourth case draws its tables from tests/test_gpu_prune_adversarial.py (cancelling rows, equal alpha, ...).
Every option form and split count must return the lists of the plain scan (`topk_prune` = 0, one dish range) bit for bit; a
few users per case are checked against the float64 restatement, tie order included.  Test infrastructure: seeds come from
the clock unless given; tests/test_gpu_soaks.py runs 20 fixed-seed cases under pytest.
Usage on the GPU box: python tests/soak_topk_plan.py [cases] [first seed]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from foodrec_amd import ScoringEngine
from oracle import m2d_oracle as oracle
from helpers import TOL, assert_scores_close
from test_gpu_prune_adversarial import adversarial_tables

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())      # [cases] [first seed]
print("seed0", seed0)
for it in range(n):
    rng = np.random.default_rng(seed0 + it)
    C = 4
    E = int(rng.choice([64, 64, 128, 32, 200, 16]))
    U = int(rng.integers(300, 40000)); I = int(rng.integers(200, 30000)); k = int(rng.choice([10, 10, 1, 5, 16]))
    serving = it % 3 == 2                                                 # a few users over a larger catalogue: many dish ranges
    if serving:
        U = int(rng.integers(10, 4000)); I = int(rng.integers(20000, 150000))
    s = 1.0 / np.sqrt(E)
    PM = (rng.standard_normal((U, C + 1, E)) * s).astype(np.float32)
    RE = (rng.standard_normal((I, E)) * s).astype(np.float32)
    CE = (rng.standard_normal((C, E)) * s).astype(np.float32)
    PM[:, 1:] *= float(rng.choice([1.0, 1.0, 0.05, 6.0, 30.0]))          # how much the bounds can prune
    style = int(rng.integers(0, 4))
    if style == 1:                                                        # coarse values: many exact ties
        PM, RE, CE = np.round(PM * 8) / 8, np.round(RE * 8) / 8, np.round(CE * 8) / 8
    pat = rng.integers(1, 16, I) if rng.integers(0, 3) else rng.choice([1, 3, 15], I)      # all patterns, or three
    cats = ((pat[:, None] >> np.arange(C)[None, :]) & 1).astype(np.float32)
    if style == 2:                                                        # copies of dishes: exact ties across the catalogue
        d = min(I // 2, 64)
        RE[I - d:] = RE[:d]; cats[I - d:] = cats[:d]
    if rng.integers(0, 2):
        cats[rng.choice(I, min(I, 7), replace=False)] = 0                 # empty masks: 0 / 0 = NaN, ranked last
    if style == 3:
        PM[rng.integers(0, U, 5)] = 0.0                                   # users whose every score ties inside a pattern
    adv = None
    if it % 4 == 1:                                                       # tables built against the pruning bounds' rounding margins
        adv = str(rng.choice(["anti", "anti_alpha0", "anti_equal_alpha", "equal_alpha", "anti_mixed", "same", "hc_cancel", "tiny_low"]))
        PM, RE, CE, cats = adversarial_tables(adv, E, U, I, seed0 + it, eps=float(rng.choice([1e-3, 1e-4, 1e-5, 1e-6])),
                                              low_scale=float(rng.choice([1.0, 6.0, 30.0])),
                                              pats_per_dish=[3, 12, 15] if rng.integers(0, 2) else None)
        style = 10
    x3 = int(rng.integers(0, 2)) if E in (64, 128) else 0
    coef = [0.99, 0.99, 0.9, 0.5, 0.0, 1.25, 1.0, 0.99][(seed0 + it) % 8]    # --high_level_score_coefficient (Train_recommender.py:61-62; no draw: the seeds' tables stay what they were)
    eng = ScoringEngine(PM.astype(np.float32), RE.astype(np.float32), CE.astype(np.float32), coef=coef); eng.set_dish_categories(cats)
    eng.set_option("topk_bf16x3", x3)
    hi_first = bool(x3 and E in (64, 128) and (seed0 + it) % 3 == 0)     # "topk_form" 3: the hi x hi first form of the pipelined kernel, forced
    eng.set_option("topk_form", 3 if hi_first else 0)
    nU = int(rng.integers(1, min(U, 2100) + 1)) if serving else int(rng.integers(257, min(U, 30000) + 1))
    users = rng.integers(0, U, nU).astype(np.int32)
    du = torch.as_tensor(users, device="cuda")
    eng.set_option("topk_prune", 0); eng.set_option("variant", 101)
    s0, i0 = eng.topk_users(du, k); eng.check()
    s0, i0 = s0.cpu().numpy(), i0.cpu().numpy()
    kern = eng.last_kernel()
    rep0 = eng.get_option("topk_repaired")
    forms = [(1, 0), (5, 0), (int(rng.choice([2, 4, 7])), 0), (1, 100 + int(rng.integers(2, 40))), (0, 0)]      # 7: dish ranges keep their thresholds apart
    for prune, var in forms:
        eng.set_option("topk_prune", prune); eng.set_option("variant", var)
        s1, i1 = eng.topk_users(du, k); eng.check()
        assert eng.last_kernel() == kern
        s1, i1 = s1.cpu().numpy(), i1.cpu().numpy()
        bad = np.flatnonzero((i1 != i0).any(1) | ~((s1 == s0) | (np.isnan(s1) & np.isnan(s0))).all(1))
        assert bad.size == 0, ("case %d seed %d" % (it, seed0 + it), kern, E, U, I, k, nU, prune, var, "style", style, "repaired", rep0, eng.get_option("topk_repaired"), bad.size, bad[:5], i1[bad[:2]], i0[bad[:2]], s1[bad[:2]], s0[bad[:2]])
    if E in (64, 128) and not (E == 128 and k > 10):                       # index-exact lists: the other dtype's kernel returns the same ids
        eng.set_option("topk_bf16x3", 1 - x3); eng.set_option("topk_prune", 1); eng.set_option("variant", 0)
        s2, i2 = eng.topk_users(du, k); eng.check()
        i2 = i2.cpu().numpy()
        bad = np.flatnonzero((i2 != i0).any(1))
        assert bad.size == 0, ("case %d seed %d" % (it, seed0 + it), "split-bf16 and exact-f32 ids differ", E, U, I, k, nU, "style", style, adv, bad.size,
                               bad[:4], i2[bad[:2]], i0[bad[:2]], s2.cpu().numpy()[bad[:2]], s0[bad[:2]])
        eng.set_option("topk_bf16x3", x3)
    all_items = np.arange(I); nv = min(k, I)
    PMf, REf, CEf = PM.astype(np.float32), RE.astype(np.float32), CE.astype(np.float32)
    for r in rng.integers(0, nU, 6):
        ref = oracle.inference_f64(PMf, REf, CEf, np.full(I, users[r]), all_items, cats, coef)
        g = i0[r, :nv]
        assert np.all(g >= 0) and np.all(g < I) and len(set(g.tolist())) == nv, (it, g)
        assert_scores_close(s0[r, :nv], ref[g], what="case %d seed %d" % (it, seed0 + it))
        key = np.where(np.isnan(s0[r, :nv]), -np.inf, s0[r, :nv])
        assert np.all(key[:-1] >= key[1:]), it
        rest = np.delete(np.where(np.isnan(ref), -np.inf, ref), g)
        if rest.size and np.isfinite(key[nv - 1]):
            assert rest.max() <= key[nv - 1] + TOL * max(1.0, abs(key[nv - 1])), (it, rest.max(), key[nv - 1])
        for a in range(nv - 1):
            if s0[r, a] == s0[r, a + 1]:
                assert g[a] < g[a + 1], (it, "tie order inside the list", g, s0[r])
        if style in (1, 2, 3) and np.isfinite(key[nv - 1]):                # exact ties at the list's end: the lower id is the one kept
            ref32 = ref.astype(np.float32)
            tied_out = [d for d in np.flatnonzero(ref32 == ref32[g[nv - 1]]) if d not in set(g.tolist())]
            if x3 == 0 and tied_out and (ref32[g] == ref32[g[nv - 1]]).any():
                pass                                                       # (f64 -> f32 equality is not the kernel's: informative only)
    print("ok", it, kern, "E%d U%d I%d k%d nU%d style%d %s coef %s repaired %d" % (E, U, I, k, nU, style, adv or "", coef, eng.get_option("topk_repaired")), flush=True)
    eng.close()
print("all", n, "cases agree")
