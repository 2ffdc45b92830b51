#!/usr/bin/env python3
"""Profiling driver for the two MFMA kernels (catalogue top-k, MLP head): a few launches each at bench sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import foodrec_amd

which = sys.argv[1] if len(sys.argv) > 1 else "both"
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
C = 4
def tables(U, I, E):
    s = E ** -0.5
    PM = torch.randn((U, C + 1, E), generator=g, device=dev) * s
    RE = torch.randn((I, E), generator=g, device=dev) * s
    CE = torch.randn((C, E), generator=g, device=dev) * s
    pat = torch.randint(1, 16, (I,), generator=g, device=dev, dtype=torch.int32)
    cats = ((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).float()
    return PM, RE, CE, cats
if which in ("both", "topk"):
    U, I, E = 200_000, 100_000, 64
    PM, RE, CE, cats = tables(U, I, E)
    eng = foodrec_amd.ScoringEngine(PM, RE, CE); eng.set_dish_categories(cats)
    users = torch.randperm(U, generator=g, device=dev)[:65536].to(torch.int32)
    for _ in range(3):
        eng.topk_users(users, 10)
    eng.check()
if which in ("both", "mlp"):
    U, I, E = 200_000, 100_000, 128
    PM, RE, CE, cats = tables(U, I, E)
    eng = foodrec_amd.ScoringEngine(PM, RE, CE); eng.set_dish_categories(cats)
    K = (C + 1) * E
    rn = lambda *sh: torch.randn(sh, generator=g, device=dev)
    eng.set_mlp_head(rn(K, 256) / K ** 0.5, rn(256) * 0.1, rn(256, 64) / 16, rn(64) * 0.1, rn(64) / 8, 0.0)
    B = 1 << 20
    users = torch.randint(0, U, (B,), generator=g, device=dev, dtype=torch.int32)
    items = torch.randint(0, I, (B,), generator=g, device=dev, dtype=torch.int32)
    for _ in range(3):
        eng.score_pairs_mlp(users, items)
    eng.check()
print("done")
