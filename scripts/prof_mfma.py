#!/usr/bin/env python3
"""Profiling driver for the two MFMA kernels at the shapes bench.py quotes them on, warm: 5 untimed launches, then 10
timed ones (HIP events), each kernel.
  retrieval : 65 536 users of a 1 M-user table x 100 k dishes, E = 64, k = 10   (bench.py's `catalogue_topk` leg)
  MLP head  : 4 M pairs, 1 M users x 100 k dishes, E = 128                      (bench.py --workload mlp --embed 128)
Writes gpurun_out/prof_<tag>/mfma_bench.json (argv[2]) with the event times and the roofline fractions by bench.py's
formulas; scripts/summarize_rocprof.py recomputes the same fractions from the kernel trace's last 10 dispatches."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import foodrec_amd

which = sys.argv[1] if len(sys.argv) > 1 else "both"
out_path = sys.argv[2] if len(sys.argv) > 2 else None
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
C, WARM, TIMED = 4, 5, 10
res = {}


def tables(U, I, E):
    s = E ** -0.5
    PM = torch.randn((U, C + 1, E), generator=g, device=dev) * s
    RE = torch.randn((I, E), generator=g, device=dev) * s
    CE = torch.randn((C, E), generator=g, device=dev) * s
    pat = torch.randint(1, 16, (I,), generator=g, device=dev, dtype=torch.int32)
    cats = ((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).float()
    return PM, RE, CE, cats


def repaired(eng):
    try:
        return eng.get_option("topk_repaired")
    except Exception:                                        # noqa: BLE001  (a library without the counter)
        return None


launches = {}                                                # scan-kernel launches so far, per kernel name (for the trace's slices)


def timed(fn):
    for _ in range(WARM):
        fn()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(TIMED + 1)]
    for i in range(TIMED):
        evs[i].record()
        fn()
    evs[TIMED].record()
    torch.cuda.synchronize()
    per = [evs[i].elapsed_time(evs[i + 1]) for i in range(TIMED)]
    return sum(per) / len(per), sorted(per)[len(per) // 2]


if which in ("both", "topk"):
    U, I, E, n = 1_000_000, 100_000, 64, 65536
    PM, RE, CE, cats = tables(U, I, E)
    eng = foodrec_amd.ScoringEngine(PM, RE, CE); eng.set_dish_categories(cats)
    users = torch.randperm(U, generator=g, device=dev)[:n].to(torch.int32)
    eng.topk_users(users[:1024], 10)                          # builds the retrieval tables
    for name, x3, prune in (("topk_bf16x3", 1, 1), ("topk_bf16x3_every_tile", 1, 0), ("topk_f32", 0, 1), ("topk_f32_every_tile", 0, 0)):
        eng.set_option("topk_bf16x3", x3)
        eng.set_option("topk_prune", prune)
        avg, med = timed(lambda: eng.topk_users(users, 10))
        eng.check()
        bu = eng.get_option("topk_block_users")              # 256, or 128: another instantiation of the scan kernel, its own trace name
        kn = (eng.last_kernel(), bu)
        first = launches.get(kn, 1 if (x3 and bu == 256) else 0) + WARM      # (the table-building call ran the default kernel once)
        launches[kn] = first + TIMED
        # flops of the tiles the blocks stepped through (a block's 256 or 128 user lanes x 32 dishes each; 3 MFMA passes in split bf16)
        scanned, full = eng.get_option("topk_tiles_scanned"), eng.get_option("topk_tiles_full")
        flop = 2.0 * E * (3 if x3 else 1) * eng.get_option("topk_block_users") * 32 * scanned
        res[name] = {
            "kernel": eng.last_kernel(), "users": n, "dishes": I, "embed_size": E, "launches": TIMED, "warmup": WARM,
            "event_avg_ms": avg, "event_median_ms": med, "executed_flop_per_launch": flop,
            "frac_of_peak": flop / avg / 1e9 / (2500.0 if x3 else 157.3), "peak_TFLOPs": 2500.0 if x3 else 157.3,
            "tiles_scanned": scanned, "tiles_without_pruning": full, "pairs_per_s": n * I / avg * 1e3,
            "scan_kernel_dispatches": [first, first + TIMED], "block_users": bu,
            "what": ("event time = the whole call (plan, sort, scan kernel, merge, tie repair); the trace's kernel time is the scan "
                     "kernel alone"),
            "repaired_users": repaired(eng)}
    eng.set_option("topk_prune", 1)
    del eng, PM
if which in ("both", "mlp"):
    U, I, E, B = 1_000_000, 100_000, 128, 1 << 22
    PM, RE, CE, cats = tables(U, I, E)
    eng = foodrec_amd.ScoringEngine(PM, RE, CE); eng.set_dish_categories(cats)
    K = (C + 1) * E
    rn = lambda *sh: torch.randn(sh, generator=g, device=dev)
    eng.set_mlp_head(rn(K, 256) / K ** 0.5, rn(256) * 0.1, rn(256, 64) / 16, rn(64) * 0.1, rn(64) / 8, 0.0)
    users = torch.randint(0, U, (B,), generator=g, device=dev, dtype=torch.int32)
    items = torch.randint(0, I, (B,), generator=g, device=dev, dtype=torch.int32)
    out = torch.empty(B, device=dev)
    avg, med = timed(lambda: eng.score_pairs_mlp(users, items, out=out))
    eng.check()
    act = float((cats[items.long()] != 0).sum(1).float().mean().item())
    Ka = (1.0 + act) * E
    flop = 3.0 * 2.0 * (Ka * 256 + 256 * 64) * B
    res["mlp_bf16x3"] = {"kernel": eng.last_kernel(), "pairs": B, "users": U, "dishes": I, "embed_size": E, "launches": TIMED,
                         "warmup": WARM, "event_avg_ms": avg, "event_median_ms": med, "mean_active_categories": act,
                         "executed_flop_per_launch": flop, "frac_of_peak": flop / avg / 1e9 / 2500.0, "peak_TFLOPs": 2500.0,
                         "hbm_algorithmic_GBps": (2 * Ka * 4 + 12) * B / avg / 1e6}
print(json.dumps(res))
if out_path:
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
