#!/usr/bin/env python3
"""m2d_score_pairs_host per-call time against batch size, for each host_zero_copy setting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from foodrec_amd import ScoringEngine

U, I, C, E = 64657, 4548, 4, 64
rng = np.random.default_rng(0)
eng = ScoringEngine((rng.standard_normal((U, C + 1, E)) / 8).astype(np.float32), (rng.standard_normal((I, E)) / 8).astype(np.float32),
                    (rng.standard_normal((C, E)) / 8).astype(np.float32))
for B in (51, 4096, 8192, 16384, 32768, 65536):
    users = rng.integers(0, U, B).astype(np.int32); items = rng.integers(0, I, B).astype(np.int32)
    cats = rng.integers(0, 2, (B, C)).astype(np.float32); cats[:, 0] = 1
    line = "B=%5d:" % B
    for z in (0, 1, 2):
        eng.set_option("host_zero_copy", z)
        for _ in range(200): eng.score_pairs_host(users, items, cats)
        t0 = time.perf_counter()
        for _ in range(2000): eng.score_pairs_host(users, items, cats)
        line += "  zero_copy=%d %.1f us" % (z, (time.perf_counter() - t0) / 2000 * 1e6)
    print(line)
