#!/usr/bin/env python3
"""m2d_rank_candidates (the evaluator's launch) at the reference's sizes: 64 657 users x 51 candidates, top-10."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from foodrec_amd import ScoringEngine

dev = torch.device("cuda")
U, I, C, L, k = 64657, 4548, 4, 51, 10
g = torch.Generator(device=dev); g.manual_seed(1)
for E in (32, 64, 128, 200):
    PM = torch.randn((U, C + 1, E), generator=g, device=dev) * E ** -0.5
    RE = torch.randn((I, E), generator=g, device=dev) * E ** -0.5
    CE = torch.randn((C, E), generator=g, device=dev) * E ** -0.5
    pat = torch.randint(1, 2 ** C, (I,), generator=g, device=dev, dtype=torch.int32)
    eng = ScoringEngine(PM, RE, CE)
    eng.set_dish_categories(((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).float())
    users = torch.arange(U, device=dev, dtype=torch.int32)
    items = torch.randint(0, I, (U, L), generator=g, device=dev, dtype=torch.int32)
    for _ in range(3): eng.rank_candidates(users, items, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eng.rank_candidates(users, items, k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    byt = U * ((C + 1) * E * 4) + U * L * (E * 4 + 4)          # a user's block once, every candidate's row
    print("E=%3d  %-24s %.3f ms  %.2f G pairs/s  %.2f TB/s (user block once + candidate rows)" % (E, eng.last_kernel(), dt * 1e3, U * L / dt / 1e9, byt / dt / 1e12), flush=True)
