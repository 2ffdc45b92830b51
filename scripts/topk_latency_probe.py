#!/usr/bin/env python3
"""Latency of m2d_topk_users for small user batches (serving): 100k dishes, E=64, k=10."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import foodrec_amd
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
C, E, U, I = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 200_000, int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
s = E ** -0.5
PM = torch.randn((U, C + 1, E), generator=g, device=dev) * s
RE = torch.randn((I, E), generator=g, device=dev) * s
CE = torch.randn((C, E), generator=g, device=dev) * s
pat = torch.randint(1, 16, (I,), generator=g, device=dev, dtype=torch.int32)
cats = ((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).float()
eng = foodrec_amd.ScoringEngine(PM, RE, CE); eng.set_dish_categories(cats)
for kv in sys.argv[3:]:                                     # engine options, name=value
    eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for n in (1, 8, 32, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
    users = torch.randperm(U, generator=g, device=dev)[:n].to(torch.int32)
    for _ in range(3):
        eng.topk_users(users, 10)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    for i in range(20):
        evs[i].record(); eng.topk_users(users, 10)
    evs[20].record(); torch.cuda.synchronize()
    ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(20))[10]
    t0 = time.perf_counter()
    for _ in range(20):
        s_, i_ = eng.topk_users(users, 10); i_.cpu()
    wall = (time.perf_counter() - t0) / 20 * 1e3
    print("%5d users: device %.3f ms, call + D2H %.3f ms (%s), %.1f G pairs/s" % (n, ms, wall, eng.last_kernel(), n * I / ms / 1e6))
