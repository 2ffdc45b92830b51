#!/usr/bin/env python3
"""PCIe-inclusive rate: Model.predict-style host call (numpy in, numpy out) at large batches, config-2 tables."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from foodrec_amd import ScoringEngine

dev = torch.device("cuda")
U, I, C, E = 1_000_000, 100_000, 4, 64
g = torch.Generator(device=dev); g.manual_seed(1)
eng = ScoringEngine(torch.randn((U, C + 1, E), generator=g, device=dev) / 8, torch.randn((I, E), generator=g, device=dev) / 8,
                    torch.randn((C, E), generator=g, device=dev) / 8)
rng = np.random.default_rng(0)
for B in (1 << 16, 1 << 18, 1 << 19, 1 << 20, 1 << 22):
    users = rng.integers(0, U, B).astype(np.int32); items = rng.integers(0, I, B).astype(np.int32)
    cats = rng.integers(0, 2, (B, C)).astype(np.float32); cats[:, 0] = 1
    for _ in range(2): eng.score_pairs_host(users, items, cats)
    t0 = time.perf_counter()
    n = 5
    for _ in range(n): eng.score_pairs_host(users, items, cats)
    dt = (time.perf_counter() - t0) / n
    print("B=%8d  %.3f ms per call  %.3f G pairs/s  (%.1f GB/s of ids + masks + scores)" % (B, dt * 1e3, B / dt / 1e9, B * 28 / dt / 1e9), flush=True)
