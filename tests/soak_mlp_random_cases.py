#!/usr/bin/env python3
"""One-off randomized soak of the MLP head kernels against the float64 restatement (test infrastructure, not collected
by pytest: seeds come from the clock).  Usage on the GPU box: python tests/soak_mlp_random_cases.py [cases]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from foodrec_amd import ScoringEngine
from oracle import m2d_oracle as oracle
from helpers import assert_scores_close

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())      # [cases] [first seed]
print("seed0", seed0)
for it in range(n):
    rng = np.random.default_rng(seed0 + it)
    C, E = [(4, 64), (4, 128), (5, 32), (2, 64), (4, 256)][rng.integers(0, 5)]
    B = int(rng.integers(1, 70000)); U = int(rng.integers(1, 300)); I = int(rng.integers(1, 200))
    K = (C + 1) * E; s = 1.0 / np.sqrt(E)
    PM = (rng.standard_normal((U, C + 1, E)) * s).astype(np.float32)
    RE = (rng.standard_normal((I, E)) * s).astype(np.float32)
    CE = (rng.standard_normal((C, E)) * s).astype(np.float32)
    users = rng.integers(0, U, B).astype(np.int32); items = rng.integers(0, I, B).astype(np.int32)
    dc = rng.integers(0, 2, (I, C)).astype(np.float32)
    if rng.integers(0, 2): dc *= rng.uniform(0.1, 3.0, (I, C)).astype(np.float32)
    head = ((rng.standard_normal((K, 256)) * 4 / np.sqrt(K)).astype(np.float32), (rng.standard_normal(256) * 0.1).astype(np.float32),
            (rng.standard_normal((256, 64)) / 4).astype(np.float32), (rng.standard_normal(64) * 0.1).astype(np.float32),
            (rng.standard_normal(64) / 2).astype(np.float32), 0.125)
    coef = [0.99, 0.9, 0.5, 0.0, 1.25, 1.0][(seed0 + it) % 6]            # --high_level_score_coefficient (no draw: the seeds' tables stay what they were)
    eng = ScoringEngine(PM, RE, CE, coef=coef); eng.set_dish_categories(dc); eng.set_mlp_head(*head)
    eng.set_option("skip_masked", int(rng.integers(0, 2)))
    eng.set_option("mlp_form", int(rng.integers(0, 3) == 0))
    got = eng.score_pairs_mlp(torch.as_tensor(users, device="cuda"), torch.as_tensor(items, device="cuda")); eng.check()
    pick = rng.integers(0, B, min(B, 3000))
    ref = oracle.inference_mlp(PM, RE, CE, dc, *head, users[pick], items[pick], coef=coef)
    assert_scores_close(got.cpu().numpy()[pick], ref, what="case %d seed %d C%d E%d B%d" % (it, seed0 + it, C, E, B))
    if it % 10 == 0: print("ok", it, C, E, B, eng.last_kernel(), flush=True)
print("all", n, "cases agree")
