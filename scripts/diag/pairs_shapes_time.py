#!/usr/bin/env python3
"""Pair-scoring rate for shapes off the benchmark's (C = 4, E = 64): which kernel runs and how fast."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from foodrec_amd import ScoringEngine

dev = torch.device("cuda")
U, I, B = 1_000_000, 100_000, 1 << 22
g = torch.Generator(device=dev); g.manual_seed(1)
for C, E in ((4, 64), (3, 64), (6, 64), (5, 32), (4, 200), (4, 36), (4, 66), (2, 128), (8, 64)):
    PM = torch.randn((U, C + 1, E), generator=g, device=dev) * E ** -0.5
    RE = torch.randn((I, E), generator=g, device=dev) * E ** -0.5
    CE = torch.randn((C, E), generator=g, device=dev) * E ** -0.5
    users = torch.randint(0, U, (B,), generator=g, device=dev, dtype=torch.int32)
    items = torch.randint(0, I, (B,), generator=g, device=dev, dtype=torch.int32)
    pat = torch.randint(1, 2 ** C, (B,), generator=g, device=dev, dtype=torch.int32)
    cats = ((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).float().contiguous()
    eng = ScoringEngine(PM, RE, CE)
    out = torch.empty(B, device=dev)
    for _ in range(3): eng.score_pairs(users, items, cats, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eng.score_pairs(users, items, cats, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    act = cats.sum(1).mean().item()
    byt = (2 + act) * E * 4 + C * 4 + 12
    print("C=%d E=%3d  %-26s %6.2f G pairs/s  %5.2f TB/s on (2 + %.2f) rows" % (C, E, eng.last_kernel(), B / dt / 1e9, B * byt / dt / 1e12, act), flush=True)
    del eng, PM
