#!/usr/bin/env python3
"""bench.py -- scored (user, dish) pairs/sec of the Market2Dish scoring path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path (``m2d_score_pairs``: user-block gather + dish-row gather + masked dots + blend,
Model_Recommender.py:56-97) over one batch of B synthetic pairs whose inputs are already in HBM.  At N = 1 the
workload is BASELINE.json configs[1] (1 M users x 100 k dishes, E = 64).  At N > 1 every rank owns a user-range shard
of that size (weak scaling) and scores B pairs whose users fall in its shard; there is no data-path
collective for pair scoring.

Rank 0 prints ONE JSON line (and writes it to --out) with `roofline` (HBM, from HIP events around every launch on the
stream the kernel runs on) and `cpu_baseline` (the CPU restatement of the reference graph timed on this box's host
cores; TF itself is unavailable).  Legs outside the timed region (benchlib/) add, as nested objects AND as top-level
scalars: `survey_8d_*` (SURVEY.md 8d's byte count on a run that fetches every row), `hbm_only_*`, `stream_probe_GBps`,
`cfg2_mlp_*` (BASELINE configs[2], the MFMA head at E = 128), `cfg4_topk_*` (configs[4], one round of the E = 128
retrieval), `topk_path_*` / `scaling_path` (configs[3], the user-sharded top-k path north_star's ">= 6x at 8 GPUs"
speaks of), `cfg1_ingredients_*`, `catalogue_topk`, `evaluator`.  `--config 3|4` makes the sharded top-k path the
timed step itself.  An N > 1 line carries `world`: ranks_seen, distinct_devices, backend, RCCL version.

`--gpus N` with N > 1 and no launcher environment: this process starts the N ranks itself (the command above, as a
child, BEFORE importing torch or touching a GPU) and exits with the child's status.

Exit status: 3 when an in-run parity check of a timed kernel against the CPU restatement fails; 4 when a leg after
the timed region did not return within --side-timeout (the headline line is still printed; `side_legs` names the leg
and rank in flight).
"""
from __future__ import annotations

import json
import os
import subprocess  # noqa: F401  (cli.launch_ranks runs the child through it; tests patch it by this name)
import sys
import threading

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchlib import cli, line as linelib                                     # noqa: E402
from benchlib.cli import parse                                                # noqa: E402,F401
from benchlib.common import (HBM_PEAK_GBS, INFINITY_CACHE_BYTES, PARITY_TOL,   # noqa: E402,F401
                             algorithmic_bytes_per_pair, make_inputs, time_steps, usable_cores)
from benchlib.sharded import (default_round_users, routed_pairs_leg,          # noqa: E402,F401
                              scaling_path_block, sharded_all_users_leg, sharded_topk_leg, world_identity)

NOTES_PAIRS = ("uniform random (user, dish) pairs with per-pair category masks; reference forward "
               "Model_Recommender.py:56-97 (ingredient table / MLP head: build-defined extensions, not in this step)")
NOTES_SKIP = ("; Personal_Memory rows of categories with mask weight 0 are not fetched -- the reference graph "
              "multiplies them by 0 (option skip_masked = 1, same scores to the bit)")
TRAFFIC_KIND = ("L2<->fabric bytes per launch (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE = TCC_EA0 request counters); they "
                "INCLUDE Infinity-Cache hits, so this bounds HBM bytes from above; a COUNTER READING OF ANOTHER RUN of "
                "this command (separate --pmc passes, scripts/profile_gpu.sh), read from the committed file and kept "
                "only while this box's streaming-read probe is within 10 % of the profiled box's (bytes per launch are "
                "a property of the kernel and its inputs, not of the box's speed; the probe guards against another "
                "part)")


def launch_ranks(a):
    return cli.launch_ranks(a, __file__)


def committed_traffic(E, B, U, I, skip):
    """profiles/traffic.json: the PMC reading of this command taken by scripts/profile_gpu.sh (another run)."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        rec = json.load(open(tpath)).get("E%d_B%d_U%d_I%d%s" % (E, B, U, I, "_skip" if skip else ""), {})
    except (OSError, ValueError):
        return None, None
    return rec.get("fabric_bytes_per_launch", rec.get("hbm_bytes_per_launch")), rec.get("stream_probe_GBps")


def pair_roofline_text(bpp, bpp_survey, skip):
    if not skip:
        return "roofline.frac follows SURVEY.md 8d: (C + 2) x E x 4 + C x 4 + 12 bytes per pair"
    return ("roofline.frac prices the bytes the timed kernel has to move: (2 + active categories) x E x 4 + C x 4 + 12 "
            "per pair, %.1f B on this batch -- the user block's rows of categories the dish does not have are "
            "multiplied by 0 in Model_Recommender.py:82 and are not fetched.  SURVEY.md 8d's formula charges all C "
            "rows (%d B per pair); it is published from a run that fetches them all: roofline.survey_8d_frac = %d B x "
            "pairs / roofline.survey_8d_ms / peak" % (bpp, bpp_survey, bpp_survey))


def topk_workload_text(a, world, U, I, E, round_cfg, n_tk):
    if a.config is not None:
        wl = ("BASELINE configs[%d]: %d users over %d GPU(s) x %d replicated dishes, E=%d: top-10 for EVERY user in "
              "rounds of %d + ONE all-gather" % (a.config, world * U, world, I, E, round_cfg))
        notes = ("%d users per GPU; all-gather of [shard,10] x (f32 score, i32 id), %d bytes per rank; build-defined "
                 "generalisation of evaluate.py:39-63" % (U, U * 80))
    else:
        wl = ("BASELINE configs[3]/[4] retrieval: full-catalogue top-%d for %d users per GPU over %d replicated "
              "dishes, E=%d, then all-gather" % (a.topk_k, n_tk, I, E))
        notes = ("users from this GPU's %d-user shard; all-gather of [users,%d] x (f32 score, i32 id); build-defined "
                 "generalisation of evaluate.py:39-63" % (U, a.topk_k))
    if a.topk_weighted_masks:
        wl += " -- WEIGHTED category masks"
        notes += ("; any float is legal placeholder input (Model_Recommender.py:32): no pattern grouping, the dense "
                  "exact-f32 kernel contracts over (C + 1) E")
    if a.topk_with_ingredients:
        wl += " -- WITH the ingredient table (%d rows)" % a.ingredients
        notes += "; build-defined ingredient table, 1-20 ingredients per dish, retrieval over [H[d] | RE[d]] rows"
    return wl, notes


def main():
    a = parse()
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # started by torch.distributed.run
    if a.gpus > 1 and not launched:
        sys.exit(launch_ranks(a))                                        # before torch / the GPU is touched
    if a.dry_run:
        return cli.dry_run(a)
    import torch
    import torch.distributed as dist

    from benchlib import baselines, configs, evaluator, mlp as mlplib, pairs as pairlib, topk as topklib, train
    from benchlib.common import random_masks

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = launched
    # M2D_BENCH_REHEARSE_ONE_GPU=1: every rank on cuda:0, torch.distributed over gloo (device tensors) -- a FUNCTIONAL
    # run of the N > 1 legs on a one-GPU box (RCCL wants a GPU per rank); the line says so, its timings mean nothing
    rehearse = use_dist and os.environ.get("M2D_BENCH_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local = 0
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: foodrec_amd has no CPU fallback")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist_or_none = dist if use_dist else None

    if not os.path.exists(os.path.join(ROOT, "foodrec_amd", "libm2d.so")):      # clean checkout: hipcc, ~15 s
        if local == 0:
            import __graft_entry__
            __graft_entry__.build()
        if use_dist:
            dist.barrier()
    import foodrec_amd
    if a.workload == "train":
        if rank == 0:
            cli.emit(train.train_workload(a, torch, foodrec_amd, dev), a.out)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    if a.config is not None:                              # BASELINE configs[3] / [4]: the sharded top-k path is timed
        a.workload = "topk"
        a.users = -(-10_000_000 // world)
        a.dishes = 1_000_000
        a.embed = 64 if a.config == 3 else 128
    C, E, U, I, B = 4, a.embed, a.users, a.dishes, a.pairs
    user_base = rank * U
    PM, RE, CE, users, items, cats = make_inputs(torch, dev, U, I, C, E, B, 20260101 + 2 + rank, user_base)
    if a.unique_users:
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        users = (torch.randperm(U, generator=g, device=dev)[:B].to(torch.int32) + int(user_base)).contiguous()
    eng = foodrec_amd.ScoringEngine(PM, RE, CE, coef=0.99, device=dev, user_base=user_base)
    for kv in a.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    out = torch.empty(B, dtype=torch.float32, device=dev)
    wl = a.workload
    g = torch.Generator(device=dev)
    g.manual_seed(20260101 + 3)                                          # same on every rank: replicated tables
    mlp_head = mlp_cats = None
    if wl in ("mlp", "topk"):
        _, dcat = random_masks(torch, I, C, dev, g)
        mlp_cats = dcat
        if wl == "topk" and a.topk_weighted_masks:              # any float weight is legal input
            dcat = dcat * (0.5 + 1.5 * torch.rand((I, C), generator=g, device=dev))
        eng.set_dish_categories(dcat)
    if wl == "mlp":
        mlp_head = mlplib.synthetic_head(torch, (C + 1) * E, dev, g)
        eng.set_mlp_head(*mlp_head)
    if wl == "ingredients" or (wl == "topk" and a.topk_with_ingredients):
        pairlib.set_synthetic_ingredients(torch, eng, I, E, a.ingredients, dev, g)
    tk_users = sharded = None
    round_cfg = default_round_users(U, a.round_users)                    # users per launch of the sharded top-k path
    if wl == "topk":
        from foodrec_amd.sharding import UserShardedScorer
        n_tk = U if a.config is not None else min(a.topk_users if a.topk_users > 0 else 65536, U)
        gen = torch.Generator(device=dev).manual_seed(11 + rank)
        tk_users = (torch.randperm(U, generator=gen, device=dev)[:n_tk].to(torch.int32) + int(user_base)).contiguous()
        sharded = UserShardedScorer(eng, world * U, device=dev)

    def step():
        if wl == "pairs":
            eng.score_pairs(users, items, cats, out=out)
        elif wl == "ingredients":
            eng.score_pairs_ingredients(users, items, cats, out=out)
        elif wl == "mlp":
            eng.score_pairs_mlp(users, items, out=out)
        elif a.config is not None:                                       # every user of the shard, ONE all-gather
            sharded.topk_all_users(10, round_users=round_cfg)
        else:                                                            # retrieval: per-shard top-k, then the exchange
            sharded.topk_users_gathered(tk_users, a.topk_k)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # clock settle (disclosed in the line: config.settle): the part ramps its clocks over the first tens of ms of
    # work after an idle period -- with the driver's W = 5, K = 20 the timed launches ran 6 % slower than launches
    # 30 ... 230 of a longer run.  Untimed, before the W warmup steps, the same step on the same buffers; at most
    # --settle-ms of wall time
    settle = {"ms_budget": a.settle_ms, "launches": 0, "ms": 0.0}
    if a.settle_ms > 0:
        import time as _time
        t_s = _time.perf_counter()
        while (_time.perf_counter() - t_s) * 1e3 < a.settle_ms:
            step()
            settle["launches"] += 1
            torch.cuda.synchronize()
        settle["ms"] = (_time.perf_counter() - t_s) * 1e3
    for _ in range(a.warmup):
        step()
    eng.check()
    barrier()
    wall, per_launch_ms = time_steps(torch, eng, users, items, cats, out, a.steps, step)
    barrier()
    eng.check()
    t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())

    # what the timed region ran and produced -- read BEFORE any side leg launches another kernel or writes a buffer
    kernel_used = eng.last_kernel()
    opts_used = {k: eng.get_option(k) for k in ("prefetch", "nt_loads", "blocks_per_cu", "skip_masked")}
    Bc = min(1 << 18, B)
    timed_sample = out[:Bc].clone() if wl == "pairs" else None           # parity sample of the TIMED kernel's scores
    mlp_sample = out[:4096].clone() if wl == "mlp" else None             # same, for the head (checked in mlp_baseline)
    ident = world_identity(torch, dist_or_none, dev, world, rank)        # who took part (collective: every rank)

    survey = None
    if rank == 0 and wl == "pairs" and opts_used["skip_masked"] != 0 and not a.no_side:
        survey = pairlib.survey_8d_leg(torch, eng, users, items, cats, C, E, a.steps)

    rc = 0
    line = None
    if rank == 0:
        bpp_survey = algorithmic_bytes_per_pair(C, E)
        skip = wl in ("pairs", "ingredients") and eng.get_option("skip_masked") != 0
        mean_active = float((cats != 0).sum(1).float().mean().item()) if skip else float(C)
        bpp = algorithmic_bytes_per_pair(C, E, mean_active) if skip else bpp_survey
        avg_ms = sum(per_launch_ms) / len(per_launch_ms)
        achieved = bpp * B / (avg_ms * 1e-3) / 1e9
        traffic, traffic_probe = committed_traffic(E, B, U, I, skip)
        table_bytes = 4 * (PM.numel() + RE.numel())
        cache_resident = table_bytes <= INFINITY_CACHE_BYTES
        units = (tk_users.numel() * I) if wl == "topk" else B           # (user, dish) pairs scored per step per GPU
        line = {
            "metric": "scored (user,dish) pairs/sec", "value": world * units * a.steps / wall_max, "unit": "pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall_max / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic %d users x %d dishes per GPU, C=%d categories, "
                                   "E=%d, %d pairs per step per GPU" % (U, I, C, E, B),
                       "notes": NOTES_PAIRS + (NOTES_SKIP if skip else ""),
                       "users_per_gpu": U, "dishes": I, "categories": C, "embed_size": E, "pairs_per_step_per_gpu": B,
                       "sharding": "user-range shard per GPU, dishes replicated, no data-path collective",
                       "kernel": kernel_used, "options": opts_used,
                       "roofline_frac_is": pair_roofline_text(bpp, bpp_survey, skip)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/traffic.json" if traffic is not None else None,
                         "traffic_kind": TRAFFIC_KIND if traffic is not None else None,
                         "kernel_avg_ms": avg_ms, "algorithmic_bytes_per_pair": bpp, "pairs_per_launch": B,
                         "table_bytes": table_bytes},
            "world": ident, **cli.ident_scalars(ident),
        }
        settle["what"] = ("untimed launches of the step BEFORE the W warmup steps, so that the timed steps run at the "
                          "clocks the part holds under this load (not counted in `steps` / `warmup`; "
                          "--settle-ms 0: off)")
        line["config"]["settle"] = settle
        if survey is not None:
            line["roofline"].update(survey)
        if skip:
            line["roofline"].update({
                "mean_active_categories": mean_active, "survey_bytes_per_pair": bpp_survey,
                "bytes_model": "mask-aware: (2 + active categories) x E x 4 + C x 4 + 12 per pair, averaged over the "
                               "batch; SURVEY.md 8d's count (%d B, all C low-level rows) applied to this run's time "
                               "would read %.3f of the peak -- more bytes than were moved"
                               % (bpp_survey, bpp_survey * B / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)})

    # the headline is complete here; everything below decorates it.  Should a leg never return (a collective that
    # does not complete on some rank), rank 0 still prints the line and every rank leaves.
    in_flight = {"leg": "none"}

    def give_up():
        # a leg did not return (a collective that never completes on some rank, a hung launch): the headline is not
        # lost, but the run did NOT end cleanly -- exit status 4, and the record says which leg this rank was in
        msg = "rank %d: leg '%s' did not return within %.0f s" % (rank, in_flight["leg"], a.side_timeout)
        print("bench.py: " + msg, file=sys.stderr)
        sys.stderr.flush()
        if rank == 0:
            line["side_legs"] = {"status": "not finished: headline line only", "rank": rank,
                                 "leg_in_flight": in_flight["leg"], "timeout_s": a.side_timeout, "exit_status": 4}
            line.update(linelib.config_scalars(line))
            cli.emit(linelib.ordered(line), a.out)
        os._exit(4)
    watchdog = threading.Timer(a.side_timeout, give_up)
    watchdog.daemon = True
    watchdog.start()

    def guarded(name, fn, *args, **kw):
        """Run a side leg; an exception costs that leg, never the headline line."""
        in_flight["leg"] = name
        try:
            return fn(*args, **kw)
        except Exception as e:                                         # noqa: BLE001
            return {"error": "%s: %s" % (type(e).__name__, e)}

    # legs every rank takes part in: the user-sharded retrieval step, the owner-routed pair step, scaling_path
    topk_ag = routed = scaling = None
    side = not a.no_side and wl == "pairs"
    if side:
        if a.topk_users > 0:
            topk_ag = guarded("sharded_topk_allgather", sharded_topk_leg, torch, dist_or_none, eng, U, I, C, E, dev,
                              user_base, min(a.topk_users, U), world)
        if use_dist:
            routed = guarded("routed_pairs_alltoall", routed_pairs_leg, torch, dist, eng, U, I, C, dev, world,
                             min(B, 1 << 22))
        if a.scaling_users > 0:
            per_gpu = -(-a.scaling_users // world)
            scaling = guarded("scaling_path", scaling_path_block, torch, dist_or_none, foodrec_amd, dev, world, rank,
                              a.scaling_users, 1_000_000, 64, 10, default_round_users(per_gpu, a.round_users))
            if world == 1 and not a.no_projection and "error" not in scaling and a.scaling_users >= 8:
                from benchlib.sharded import projected_world8_block
                scaling["projected_world8"] = guarded("scaling_path.projected_world8", projected_world8_block, torch,
                                                      foodrec_amd, dev, a.scaling_users, 1_000_000, 64, 10,
                                                      scaling.get("wall_ms"))
    in_flight["leg"] = "rank-0 side measurements"

    if a.sweep and rank == 0:
        bpp_now = line["roofline"]["algorithmic_bytes_per_pair"]
        pairlib.knob_sweep(torch, eng, users, items, cats, out, bpp_now, opts_used,
                           lambda s: print(s, file=sys.stderr))

    if rank == 0:
        if cache_resident and wl in ("pairs", "ingredients"):
            # both tables stay in the 256 MiB Infinity Cache between launches: the algorithmic rate is a cache rate and
            # can exceed the HBM peak, so no HBM fraction is published for this shape
            line["roofline"].update({"bound": "cache", "peak": None, "frac": None,
                                     "note": "tables (%.0f MB) fit the 256 MiB Infinity Cache: rows are re-read "
                                             "on-die, the HBM roofline does not bound this shape"
                                             % (table_bytes / 1e6)})
        if wl == "mlp":
            K = (C + 1) * E
            line["config"]["workload"] = ("BASELINE configs[2]: synthetic %d users x %d dishes per GPU, C=%d, E=%d + "
                                          "BUILD-DEFINED 3-layer head %d->256->64->1, %d pairs per step"
                                          % (U, I, C, E, K, B))
            line["config"]["notes"] = ("head on the interaction vector: no reference counterpart, parity vs the "
                                       "build's own restatement only; uniform random pairs, masks from the resident "
                                       "dish table")
            line["roofline"], line["dtype"] = mlplib.mlp_roofline(torch, eng, kernel_used, mlp_cats, items, C, E, B,
                                                                  avg_ms)
            if not a.no_cpu_baseline and world == 1:
                in_flight["leg"] = "mlp cpu_baseline"
                cb, ok = baselines.mlp_baseline(torch, PM, RE, CE, mlp_cats, mlp_head, users, items, user_base,
                                                mlp_sample, a.cpu_seconds)
                line["cpu_baseline"] = cb
                if not ok:
                    rc = 3
        if wl == "ingredients":
            bpp_i = pairlib.ingredients_bytes_per_pair(C, E, mean_active)
            ach = bpp_i * B / (avg_ms * 1e-3) / 1e9
            line["config"]["workload"] = ("BASELINE configs[1] WITH the build-defined ingredient table: %d users x %d "
                                          "dishes x %d ingredients per GPU, 1-20 per dish, E=%d"
                                          % (U, I, a.ingredients, E))
            line["config"]["notes"] = ("high-level path from the per-dish multi-hot ingredient sum (hoisted to a "
                                       "per-table segment-sum kernel), low-level path and blend as "
                                       "Model_Recommender.py:82-96; no reference counterpart")
            line["roofline"].update({"achieved": ach, "algorithmic_bytes_per_pair": bpp_i, "traffic": None,
                                     "traffic_kind": None})
            if not cache_resident:
                line["roofline"]["frac"] = ach / HBM_PEAK_GBS
        if wl == "topk":
            wtxt, notes = topk_workload_text(a, world, U, I, E, round_cfg, tk_users.numel())
            line["config"]["workload"], line["config"]["notes"] = wtxt, notes
            roof, scanned_frac, line["dtype"] = topklib.timed_topk_roofline(eng, kernel_used, C, E, units, avg_ms,
                                                                            a.topk_with_ingredients)
            line["roofline"] = roof
            # "scored" is claimed only for the pairs that were multiplied: the tiles the blocks stepped through.  Every
            # pair of the catalogue is DECIDED (ranked or excluded by a bound) at the rate beside it.
            decided = line["value"]
            line["pairs_decided_per_s"] = decided
            line["value"] = decided * (scanned_frac if scanned_frac is not None else 1.0)
            line["pairs_multiplied_per_s"] = line["value"]
            line["value_definition"] = ("v2 (rounds 4+): pairs MULTIPLIED per second; the BENCH / profiles records of "
                                        "rounds 1-3 published what is now pairs_decided_per_s under `value`")
            line["value_is"] = ("(user, dish) pairs multiplied per second, whole job: every pair of the catalogue is "
                                "decided at pairs_decided_per_s, the share `roofline.scanned_fraction` of them by "
                                "being scored -- the others by a bound on their mask pattern's scores (DESIGN.md 4.4)")
            if sharded is not None and getattr(sharded, "last_allgather_events", None):
                ev = sharded.last_allgather_events
                line["allgather_exposed_ms"] = ev[0].elapsed_time(ev[1])
            roof["allgather_bytes_per_rank"] = tk_users.numel() * 8 * a.topk_k if use_dist else 0
            roof["repaired_users_last_launch"] = eng.get_option("topk_repaired")
        if side:
            in_flight["leg"] = "no-reuse / stream probe"
            nr, probe, hbm_only = pairlib.side_measurements(torch, eng, PM, U, I, C, E, dev, user_base)
            line["roofline"].update({"no_reuse": nr, "hbm_only": hbm_only, "stream_read_probe": probe})
            line["roofline"].update(pairlib.side_scalars(achieved, nr, probe, hbm_only))
            if traffic is not None:
                # the counter reading belongs to the box it was taken on: kept only while this box streams like it
                line["roofline"]["traffic_profiled_box_stream_probe_GBps"] = traffic_probe
                if traffic_probe is None or abs(probe["GBps"] / traffic_probe - 1.0) > 0.10:
                    line["roofline"].update({
                        "traffic": None,
                        "traffic_dropped": "this box's stream probe (%.0f GB/s) is not within 10 %% of the profiled "
                                           "box's (%s GB/s): the committed counter reading is not published for it"
                                           % (probe["GBps"],
                                              "%.0f" % traffic_probe if traffic_probe else "unrecorded")})
        if side and a.topk_users > 0:
            line["catalogue_topk"] = guarded("catalogue_topk", topklib.catalogue_topk_block, torch, eng, U, I, C, E,
                                             dev, user_base, min(a.topk_users, U))
        if side and world == 1 and not a.no_cpu_baseline:
            line["evaluator"] = guarded("evaluator", evaluator.evaluator_leg, torch, dev)
        if side:
            line["with_user_high_table"] = guarded("with_user_high_table", pairlib.user_high_leg, torch, eng, users,
                                                   items, cats, C, E)
            line["with_ingredient_table"] = guarded("with_ingredient_table", pairlib.ingredients_leg, torch, eng,
                                                    users, items, cats, I, C, E, dev, a.ingredients)
        if scaling is not None:
            if world > 1 and wl == "pairs" and "error" not in scaling:
                # a SCALE record headlines `value`, which is the collective-free pair path in weak scaling (about N x
                # by construction): say first where the path north_star scales is
                prefix = ("[topk_path_ms = %.1f is the strong-scaling path; `value` scales weakly] "
                          % (scaling.get("wall_ms") or float("nan")))
                line["config"]["workload"] = prefix + line["config"]["workload"]
            line["scaling_path"] = scaling                  # ("scaling" itself is the contract's "weak" / "strong")
            line.update(linelib.scaling_scalars(scaling))
        if topk_ag is not None:
            line["sharded_topk_allgather"] = topk_ag
        if routed is not None:
            line["routed_pairs_alltoall"] = routed
        if a.unique_users:
            line["config"]["workload"] += " [--unique-users: every user at most once per step]"
        if world == 1 and not a.no_cpu_baseline and wl == "pairs":
            in_flight["leg"] = "cpu_baseline"
            cb, ref, _ = baselines.cpu_baseline(torch, PM, RE, CE, users, items, cats, a.cpu_seconds)
            if not baselines.check_timed_sample(torch, cb, timed_sample, ref):
                rc = 3
                print("bench.py: PARITY FAILURE: timed kernel vs CPU restatement, max |d| / max(1, |ref|) = %.3e > "
                      "%.0e" % (cb["max_rel_diff_vs_gpu"], PARITY_TOL), file=sys.stderr)
            line["cpu_baseline"] = cb
        if side and world == 1 and not a.no_config_legs:
            # BASELINE configs[2] and [4] on tables of their own: the main engine's tables go first (3.9 GB are needed)
            eng.close()
            del PM, RE, CE, users, items, cats, out, eng
            torch.cuda.empty_cache()
            line["config2_mlp"] = guarded("config2_mlp", configs.config2_mlp_leg, torch, foodrec_amd, dev)
            if line["config2_mlp"].get("parity_ok") is False:
                rc = 3
                print("bench.py: PARITY FAILURE: config2 MLP head vs its float64 restatement, %.3e"
                      % line["config2_mlp"]["max_rel_vs_restatement"], file=sys.stderr)
            line["config4_topk"] = guarded("config4_topk", configs.config4_topk_leg, torch, foodrec_amd, dev)
        if rehearse:
            line["rehearsal"] = ("%d ranks share ONE GPU, collectives over gloo: a functional run of the N > 1 path "
                                 "(M2D_BENCH_REHEARSE_ONE_GPU=1); its timings and rates mean nothing" % world)
        line.update(linelib.config_scalars(line))
        watchdog.cancel()
        cli.emit(linelib.ordered(line), a.out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    watchdog.cancel()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
