#!/usr/bin/env python3
"""Create / use / close engines in a loop and watch the device's free memory (every lazily built table must be freed)."""
import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from foodrec_amd import ScoringEngine

dev = torch.device("cuda")
rng = np.random.default_rng(0)
def once(E):
    U, I, C = 3000, 2000, 4
    PM = (rng.standard_normal((U, C + 1, E)) / 8).astype(np.float32); RE = (rng.standard_normal((I, E)) / 8).astype(np.float32)
    CE = (rng.standard_normal((C, E)) / 8).astype(np.float32)
    eng = ScoringEngine(PM, RE, CE)
    dc = rng.integers(0, 2, (I, C)).astype(np.float32); dc[dc.sum(1) == 0, 0] = 1
    eng.set_dish_categories(dc)
    users = torch.as_tensor(rng.integers(0, U, 20000).astype(np.int32), device=dev)
    items = torch.as_tensor(rng.integers(0, I, 20000).astype(np.int32), device=dev)
    eng.score_pairs_bydish(users, items)
    eng.score_pairs_host(users[:51].cpu().numpy(), items[:51].cpu().numpy(), dc[items[:51].cpu().numpy()])
    eng.topk_users(users[:500].contiguous(), 10)
    K = (C + 1) * E
    eng.set_mlp_head((rng.standard_normal((K, 256)) / 30).astype(np.float32), np.zeros(256, np.float32),
                     (rng.standard_normal((256, 64)) / 16).astype(np.float32), np.zeros(64, np.float32),
                     (rng.standard_normal(64) / 8).astype(np.float32), 0.0)
    eng.score_pairs_mlp(users, items)
    eng.set_option("mlp_form", 1); eng.score_pairs_mlp(users, items)
    off = np.arange(I + 1, dtype=np.int32) * 3
    eng.set_ingredients((rng.standard_normal((500, E)) / 8).astype(np.float32), off, rng.integers(0, 500, off[-1]).astype(np.int32))
    eng.score_pairs_ingredients(users, items)
    eng.topk_users(users[:500].contiguous(), 10)
    eng.clear_ingredients()
    eng.train_begin("adam", 0.01)
    eng.train_step(users[:256], items[:256], torch.as_tensor(dc, device=dev)[items[:256].long()], torch.ones(256, device=dev))
    eng.train_end()
    gm = torch.zeros((7, C + 1, E), device=dev)
    eng.write_memory(users[:256], items[:256], torch.as_tensor(dc, device=dev)[items[:256].long()], torch.ones(256, device=dev),
                     torch.ones((256, 7), device=dev), gm, 0.01, 0.02, 0.03)
    eng.check(); eng.close()

for E in (64, 200): once(E)
gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
free0 = torch.cuda.mem_get_info()[0]
for it in range(60):
    once(64 if it % 2 else 200)
gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info()[0]
print("free before %.1f MiB, after 60 engines %.1f MiB, difference %.2f MiB" % (free0 / 2**20, free1 / 2**20, (free0 - free1) / 2**20))
sys.exit(1 if free0 - free1 > 32 * 2**20 else 0)
