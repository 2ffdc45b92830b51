"""BASELINE configs[2] and configs[4] as side legs of the default run (outside the timed region): each builds its own
engine on tables of that config's shape, times a bounded number of steps with HIP events and frees everything."""
from __future__ import annotations

import time

from .common import PARITY_TOL, make_inputs, masks_from_patterns, median, random_masks, settle, time_steps
from .mlp import mlp_roofline, synthetic_head
from .topk import timed_topk_roofline


def _cleanup(torch, dev, eng):
    eng.close()
    if torch.device(dev).type == "cuda":
        torch.cuda.empty_cache()


def config2_mlp_leg(torch, foodrec_amd, dev, users=1_000_000, dishes=100_000, E=128, pairs=1 << 21, steps=10,
                    parity_pairs=2048, settle_ms=150.0):
    """BASELINE configs[2]: 1 M users x 100 k dishes, E = 128 + the build-defined head 640 -> 256 -> 64 -> 1, `pairs`
    uniform random pairs per step.  The timed kernel's first `parity_pairs` scores are compared with the float64
    restatement of the head (the head has no reference counterpart)."""
    import numpy as np
    from oracle import m2d_oracle                         # the checker, outside the timed steps
    C = 4
    PM, RE, CE, u, d, _ = make_inputs(torch, dev, users, dishes, C, E, pairs, 20260101 + 2, 0)
    g = torch.Generator(device=dev)
    g.manual_seed(20260101 + 3)
    _, dcat = random_masks(torch, dishes, C, dev, g)
    eng = foodrec_amd.ScoringEngine(PM, RE, CE, coef=0.99, device=dev)
    try:
        eng.set_dish_categories(dcat)
        head = synthetic_head(torch, (C + 1) * E, dev, g)
        eng.set_mlp_head(*head)
        out = torch.empty(pairs, dtype=torch.float32, device=dev)

        def step():
            eng.score_pairs_mlp(u, d, out=out)
        # the leg comes behind the CPU baseline's seconds of an idle GPU: untimed launches for `settle_ms` first, as in
        # front of the headline's warmup steps (config.settle), then three more
        settle_n = settle(torch, step, settle_ms)
        time_steps(torch, eng, u, d, None, out, 3, step)
        wall, per = time_steps(torch, eng, u, d, None, out, steps, step)
        eng.check()
        kernel = eng.last_kernel()
        ms = sum(per) / len(per)
        roof, dtype = mlp_roofline(torch, eng, kernel, dcat, d, C, E, pairs, ms)
        n = parity_pairs
        ui, di = u[:n].cpu().numpy().astype(np.int64), d[:n].cpu().numpy().astype(np.int64)
        rows, drows = np.unique(ui), np.unique(di)        # only the sampled users' and dishes' rows leave the device
        pm_small = PM[torch.from_numpy(rows).to(dev)].cpu().numpy()
        dsel = torch.from_numpy(drows).to(dev)
        hd = [h.cpu().numpy() if hasattr(h, "cpu") else h for h in head]
        ref = m2d_oracle.inference_mlp(pm_small, RE[dsel].cpu().numpy(), CE.cpu().numpy(), dcat[dsel].cpu().numpy(),
                                       *hd, np.searchsorted(rows, ui), np.searchsorted(drows, di))
        got = out[:n].cpu().numpy().astype(np.float64)
        err = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
        return {"users": users, "dishes": dishes, "embed_size": E, "pairs_per_step": pairs, "steps": steps,
                "settle_launches": settle_n, "kernel": kernel, "dtype": dtype, "kernel_avg_ms": ms,
                "ms_per_step_wall": wall / steps * 1e3,
                "pairs_per_s": pairs * steps / wall, "roofline": roof, "max_rel_vs_restatement": err,
                "parity_pairs": n, "parity_tolerance": PARITY_TOL, "parity_ok": bool(err <= PARITY_TOL),
                "what": "BASELINE configs[2]: build-defined head %d->256->64->1 on the interaction vector; parity "
                        "against the build's own float64 restatement only" % ((C + 1) * E)}
    finally:
        _cleanup(torch, dev, eng)


def config4_topk_leg(torch, foodrec_amd, dev, round_users=500_000, dishes=1_000_000, E=128, k=10, reps=3):
    """BASELINE configs[4]: E = 128 full-catalogue top-10 over 1 M replicated dishes -- ONE round of `round_users`
    users (N = 1 runs 20 such rounds over its 10 M users; `--config 4` times them all)."""
    C = 4
    g = torch.Generator(device=dev)
    g.manual_seed(20260101 + 4)
    sc = E ** -0.5
    RE = torch.randn((dishes, E), generator=g, device=dev) * sc
    CE = torch.randn((C, E), generator=g, device=dev) * sc
    pat = torch.randint(1, 2 ** C, (dishes,), generator=g, device=dev, dtype=torch.int32)
    PM = torch.randn((round_users, C + 1, E), generator=g, device=dev) * sc
    eng = foodrec_amd.ScoringEngine(PM, RE, CE, coef=0.99, device=dev)
    try:
        eng.set_dish_categories(masks_from_patterns(torch, pat, C, dev))
        users = torch.arange(round_users, dtype=torch.int32, device=dev)
        s = torch.empty((round_users, k), dtype=torch.float32, device=dev)
        ids = torch.empty((round_users, k), dtype=torch.int32, device=dev)
        t0 = time.perf_counter()
        eng.topk_users_into(users, k, s, ids)             # builds the retrieval tables
        for _ in range(4):                                # ~120 ms of the same call: the clocks have ramped
            eng.topk_users_into(users, k, s, ids)
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t0
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        for i in range(reps):
            evs[i].record()
            eng.topk_users_into(users, k, s, ids)
        evs[reps].record()
        torch.cuda.synchronize()
        eng.check()
        ms = median([evs[i].elapsed_time(evs[i + 1]) for i in range(reps)])
        kernel = eng.last_kernel()
        units = round_users * dishes
        roof, scanned, dtype = timed_topk_roofline(eng, kernel, C, E, units, ms, False)
        ok = bool((ids >= 0).all().item() and (s[:, :-1] >= s[:, 1:]).all().item())
        return {"round_users": round_users, "dishes": dishes, "embed_size": E, "k": k, "reps": reps,
                "kernel": kernel, "dtype": dtype, "round_ms": ms, "users_per_s": round_users / ms * 1e3,
                "pairs_decided_per_s": units / ms * 1e3,
                "pairs_multiplied_per_s": units / ms * 1e3 * (scanned if scanned is not None else 1.0),
                "roofline": roof, "lists_sorted_and_filled": ok, "first_five_calls_s": build_s,
                "what": "BASELINE configs[4]: one round of %d users of the E = 128 retrieval over %d dishes (tables "
                        "built before the timed calls)" % (round_users, dishes)}
    finally:
        _cleanup(torch, dev, eng)
