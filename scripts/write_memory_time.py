#!/usr/bin/env python3
"""Timing of Write_Memory as a scatter-add at the driver's sizes and at a large batch.
Prints one JSON object; profiles/r02_write_memory.json is this script's output on an MI355X."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import foodrec_amd

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
out = {}
for name, (U, I, C, E, L, B) in {"reference sizes, the driver's batch (Train_recommender.py:35, :51-58)": (64657, 4548, 4, 200, 95, 256),
                                 "reference sizes, the 8-pair personal branch (:180-184)": (64657, 4548, 4, 200, 95, 8),
                                 "config-2 sizes, 65536 pairs": (1_000_000, 100_000, 4, 64, 95, 65536)}.items():
    s = E ** -0.5
    PM = torch.randn((U, C + 1, E), generator=g, device=dev) * s
    RE = torch.randn((I, E), generator=g, device=dev) * s
    CE = torch.randn((C, E), generator=g, device=dev) * s
    GM = torch.randn((L, C + 1, E), generator=g, device=dev) * s
    eng = foodrec_amd.ScoringEngine(PM, RE, CE)
    users = torch.randint(0, U, (B,), generator=g, device=dev, dtype=torch.int32)
    items = torch.randint(0, I, (B,), generator=g, device=dev, dtype=torch.int32)
    pat = torch.randint(1, 2 ** C, (B,), generator=g, device=dev, dtype=torch.int32)
    cats = ((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).float()
    sign = (torch.randint(0, 2, (B, 1), generator=g, device=dev).float() * 2 - 1)
    labels = (torch.rand((B, L), generator=g, device=dev) < 0.05).float()
    row = {}
    for what, kw in {"general fetch (GM assign only)": dict(write_pm=False, write_gm=True),
                     "personal fetch (PM assigns only)": dict(write_pm=True, write_gm=False),
                     "both": dict(write_pm=True, write_gm=True)}.items():
        step = lambda: eng.write_memory(users, items, cats, sign, labels, GM, 0.5, 0.5, 0.001, **kw)
        for _ in range(5): step()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
        torch.cuda.synchronize()
        evs[0].record()
        for i in range(40):
            step(); evs[i + 1].record()
        torch.cuda.synchronize()
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(40))
        us = per[len(per) // 2] * 1e3
        # bytes added: PM pass (C+1) E floats per pair; GM pass (C+1) E floats per (pair, label set) ~ labels per pair
        nlab = float(labels.sum(1).mean().item())
        act = float((cats != 0).sum(1).float().mean().item())            # the GM passes add U_high's row and the active categories' rows
        added = B * E * 4 * ((C + 1) * (1 if kw["write_pm"] else 0) + (1 + act) * (nlab if kw["write_gm"] else 0))
        row[what] = {"median_us": us, "pairs_per_s": B / us * 1e6, "atomic_bytes_added": added, "added_GBps": added / us / 1e3}
    eng.check()
    out[name] = {"users": U, "dishes": I, "E": E, "labels": L, "batch": B, "mean_labels_per_pair": nlab, **row}
out["note"] = ("float atomic adds, 256 contiguous bytes per wave-instruction; the part's measured ceiling for this access shape is about "
               "1.3 TB/s of added bytes (MI355X_MICROARCH.md, float atomic add); small batches are launch-bound (2 launches + the "
               "host-side reshapes of the Python wrapper)")
print(json.dumps(out, indent=1))
