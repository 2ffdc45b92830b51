#!/usr/bin/env python3
"""Latency of m2d_score_pairs_mlp for small batches (E=128, head 640->256->64->1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import foodrec_amd
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
C, E, U, I = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 200_000, 100_000
s = E ** -0.5
PM = torch.randn((U, C + 1, E), generator=g, device=dev) * s
RE = torch.randn((I, E), generator=g, device=dev) * s
CE = torch.randn((C, E), generator=g, device=dev) * s
pat = torch.randint(1, 16, (I,), generator=g, device=dev, dtype=torch.int32)
cats = ((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).float()
eng = foodrec_amd.ScoringEngine(PM, RE, CE); eng.set_dish_categories(cats)
K = (C + 1) * E
rn = lambda *sh: torch.randn(sh, generator=g, device=dev)
eng.set_mlp_head(rn(K, 256) / K ** 0.5, rn(256) * 0.1, rn(256, 64) / 16, rn(64) * 0.1, rn(64) / 8, 0.0)
for B in (1, 51, 256, 1024, 8192, 65536):
    users = torch.randint(0, U, (B,), generator=g, device=dev, dtype=torch.int32)
    items = torch.randint(0, I, (B,), generator=g, device=dev, dtype=torch.int32)
    for _ in range(3):
        eng.score_pairs_mlp(users, items)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    for i in range(20):
        evs[i].record(); eng.score_pairs_mlp(users, items)
    evs[20].record(); torch.cuda.synchronize()
    ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(20))[10]
    print("%6d pairs: %.3f ms (%s), %.1f M pairs/s" % (B, ms, eng.last_kernel(), B / ms / 1e3))
