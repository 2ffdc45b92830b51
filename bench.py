#!/usr/bin/env python3
"""bench.py -- scored (user, dish) pairs/sec of the Market2Dish scoring path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path (``m2d_score_pairs``: user-block gather + dish-row gather +
masked dots + blend, Model_Recommender.py:56-97) over one batch of B synthetic pairs whose inputs
are already in HBM.  At N = 1 the workload is BASELINE.json configs[1] (1 M users x 100 k dishes,
E = 64).  At N > 1 every rank owns a user-range shard of that size (weak scaling) and
scores B pairs whose users fall in its shard; there is no data-path collective for pair scoring.

Rank 0 prints ONE JSON line (contract in the task statement) including `roofline` (HBM, from HIP
events around every launch on the stream the kernel runs on) and `cpu_baseline` (the CPU
restatement of the reference graph timed on this box's host cores; TF itself is unavailable).

`--gpus N` with N > 1 and no launcher environment: this process starts the N ranks itself (it runs the
torch.distributed.run command above as a child BEFORE importing torch or touching a GPU) and exits with the
child's status.  At N > 1 the line also carries `sharded_topk_allgather` (the user-sharded full-catalogue
top-k + RCCL all-gather, median of 7) and `routed_pairs_alltoall` (pairs routed to the owners of their users).

Every line also carries `scaling_path`: the user-sharded top-k path north_star's ">= 6x at 8 GPUs" speaks of, at
BASELINE configs[3]'s per-GPU shape (10 M / N users x 1 M dishes, E = 64; every user of the shard in rounds of 524 288,
one all-gather of [shard, 10] x (f32, i32)) -- `--config 3|4` makes that path the timed step itself.

Exit status: 3 when the in-run parity check of the timed kernel against the CPU restatement fails; 4 when a leg after
the timed region did not return within --side-timeout (the headline line is still printed; `side_legs` names the leg
and rank in flight).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy rate
INFINITY_CACHE_BYTES = 256 << 20   # tables smaller than this never leave the die-level cache: no HBM roofline applies
PARITY_TOL = 1e-4            # north_star: scores within 1e-4 fp32, as |d| <= tol * max(1, |ref|)


def algorithmic_bytes_per_pair(C: int, E: int, mean_active=None):
    """user block + dish row + mask + two ids + score.

    `mean_active` (the batch's mean number of categories with a non-zero mask weight): the byte count of the path as
    built -- the Personal_Memory row of a category whose weight is 0 is multiplied by 0 in the reference graph
    (Model_Recommender.py:82) and is not fetched, so a pair needs U_high + `active` low-level rows, not C + 1 rows."""
    if mean_active is None:
        return (C + 2) * E * 4 + C * 4 + 12
    return (2.0 + mean_active) * E * 4 + C * 4 + 12


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--users", type=int, default=1_000_000, help="users per GPU shard")
    p.add_argument("--dishes", type=int, default=100_000)
    p.add_argument("--embed", type=int, default=64)
    p.add_argument("--pairs", type=int, default=1 << 22, help="pairs per step per GPU")
    p.add_argument("--learner", default="adam", help="workload train: adam / adagrad / rmsprop / sgd")
    p.add_argument("--workload", choices=["pairs", "ingredients", "mlp", "topk", "train"], default="pairs",
                   help="pairs = BASELINE configs[1] (reference forward, HBM-bound); ingredients = configs[1] with the "
                        "build-defined 10k-row ingredient table on the high-level path; mlp = configs[2] (build-defined "
                        "3-layer head, MFMA-bound; pass --embed 128); topk = configs[3]/[4] retrieval: full-catalogue "
                        "top-10 for --topk-users users per GPU + all-gather of the results (MFMA-bound); train = the reference's "
                        "training step (SURVEY.md 8f N4) at its own default sizes unless --users/--dishes/--embed/--pairs "
                        "are given: loss + gradients + clip + optimizer update per step, single GPU")
    p.add_argument("--ingredients", type=int, default=10_000, help="rows of the ingredient table (workload ingredients)")
    p.add_argument("--config", type=int, choices=[3, 4], default=None,
                   help="BASELINE.json configs[3] / configs[4] as the timed step: 10 M / N users per GPU x 1 M replicated dishes, "
                        "E = 64 / 128, top-10 for EVERY user of the shard in rounds of --round-users, then ONE all-gather of "
                        "[shard, 10] x (f32 score, i32 id) (100 MB per rank at N = 8)")
    p.add_argument("--round-users", type=int, default=0, help="users per retrieval launch in the sharded top-k path "
                   "(0 = the shard in the fewest even rounds of at most 524288)")
    p.add_argument("--no-projection", action="store_true", help="skip scaling_path.projected_world8 (the N = 8 per-GPU shape timed on one GPU)")
    p.add_argument("--scaling-users", type=int, default=10_000_000,
                   help="users over ALL GPUs in the scaling_path block (configs[3]: 10 M; 0 = leave the block out)")
    p.add_argument("--topk-weighted-masks", action="store_true",
                   help="--workload topk with category weights other than 0 / 1 (the placeholder is float, Model_Recommender.py:32): "
                        "the masks cannot be grouped by pattern, the dense exact-f32 kernel m2d_topk_mfma serves the call")
    p.add_argument("--topk-k", type=int, default=10, help="--workload topk: list length (k > 10 takes the 16-slot instantiations)")
    p.add_argument("--topk-with-ingredients", action="store_true",
                   help="workload topk: set the ingredient table first (retrieval over [H[d] | RE[d]] rows, E = 32 / 64)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    p.add_argument("--topk-users", type=int, default=65536, help="users in the catalogue top-k side leg (0 = skip)")
    p.add_argument("--no-side", action="store_true", help="skip the no-reuse / stream-probe side measurements")
    p.add_argument("--unique-users", action="store_true",
                   help="profiling aid: every user at most once per step (pairs <= users), no table reuse")
    p.add_argument("--sweep", action="store_true", help="also time the kernel knobs (stderr only)")
    p.add_argument("--opt", action="append", default=[], help="engine option name=value")
    p.add_argument("--side-timeout", type=float, default=420.0,
                   help="seconds the legs after the timed region may take before rank 0 prints the headline line without "
                        "them and every rank exits (a collective that never completes must not cost the line)")
    p.add_argument("--dry-run", action="store_true",
                   help="launch plumbing only (CPU, gloo): the ranks rendezvous, exchange their shard ranges and rank 0 "
                        "prints a line with value null -- nothing is scored, no GPU is touched (tests/test_bench_contract.py)")
    return p.parse_args()


def launch_ranks(a):
    """`python bench.py --gpus N` with N > 1: start one rank per GPU as fresh child processes through
    torch.distributed.run and relay their status.  Runs before torch is imported: this process never touches a GPU
    (and never exec-replaces itself)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % a.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dry_run(a):
    """--dry-run: the multi-process plumbing of this script on CPU -- rendezvous, shard ranges, one collective, the
    rank-0 line -- with no engine and no scoring."""
    import torch
    import torch.distributed as dist
    from foodrec_amd.sharding import shard_range
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if os.environ.get("M2D_BENCH_DRYRUN_FAIL_RANK") == str(rank):     # test seam: a rank that dies must fail the launcher
        sys.exit(5)
    if world > 1:
        dist.init_process_group("gloo")
    base, count = shard_range(world * a.users, world, rank)
    mine = torch.tensor([rank, base, count], dtype=torch.int64)
    allr = torch.empty(world * 3, dtype=torch.int64)
    if world > 1:
        dist.all_gather_into_tensor(allr, mine)
        dist.barrier()
    else:
        allr.copy_(mine)
    if rank == 0:
        print(json.dumps({"metric": "scored (user,dish) pairs/sec", "value": None, "unit": "pairs/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none (dry run)", "dry_run": True,
                          "config": {"workload": "launch plumbing only", "shards": allr.view(world, 3).tolist()}}))
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


def make_inputs(torch, dev, U, I, C, E, B, seed, user_base):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    s = 1.0 / (E ** 0.5)
    PM = torch.randn((U, C + 1, E), generator=g, device=dev, dtype=torch.float32) * s
    RE = torch.randn((I, E), generator=g, device=dev, dtype=torch.float32) * s
    CE = torch.randn((C, E), generator=g, device=dev, dtype=torch.float32) * s
    users = torch.randint(0, U, (B,), generator=g, device=dev, dtype=torch.int32) + int(user_base)
    items = torch.randint(0, I, (B,), generator=g, device=dev, dtype=torch.int32)
    pat = torch.randint(1, 2 ** C, (B,), generator=g, device=dev, dtype=torch.int32)   # non-empty subset
    cats = ((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).to(torch.float32)
    return PM, RE, CE, users, items, cats.contiguous()


def time_steps(torch, eng, users, items, cats, out, steps, step=None):
    """K launches; per-launch HIP-event durations (ms) on the current stream + wall seconds."""
    if step is None:
        step = lambda: eng.score_pairs(users, items, cats, out=out)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    return wall, per


def usable_cores():
    """Host cores this process may actually run on: the affinity mask capped by the cgroup CPU quota.  (A GPU box
    hands one GPU's job a share of a 256-thread host; 256 threads on that share run slower than 16.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            if parse:
                quota, period = parse(open(path).read())
            else:
                quota = open(path).read().strip()
                period = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1"):
                n = min(n, max(1, -(-int(quota) // int(period))))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


def mlp_baseline(torch, PM, RE, CE, dish_cats, head, users, items, user_base, gpu_sample, budget_s):
    """The build's float64 restatement of the 3-layer head (oracle/m2d_oracle.py::inference_mlp; the head has no reference
    counterpart) on the first pairs of the timed batch: a live parity check of the TIMED kernel's scores, and its rate
    on this box's host cores (numpy / BLAS threads as configured) beside the GPU number."""
    import numpy as np
    from oracle import m2d_oracle
    n = gpu_sample.numel()
    pm, re, ce, dc = PM.cpu().numpy(), RE.cpu().numpy(), CE.cpu().numpy(), dish_cats.cpu().numpy()
    hd = [h.cpu().numpy() if hasattr(h, "cpu") else h for h in head]
    u = (users[:n].cpu().numpy() - int(user_base)).astype(np.int64)
    d = items[:n].cpu().numpy().astype(np.int64)
    ref = m2d_oracle.inference_mlp(pm, re, ce, dc, *hd, u, d)                 # float64: the parity sample
    # the rate: the same arithmetic in float32 with the dish vectors built once (as the engine keeps them), on slices
    # of 65536 pairs of the timed batch
    Dt = m2d_oracle.dish_vectors(re, ce, dc, m2d_oracle.DEFAULT_COEF, np.float32)
    W1, b1, W2, b2, w3, b3 = [np.asarray(x, dtype=np.float32) for x in hd]
    nb = min(65536, users.numel())
    ub = (users[:nb].cpu().numpy() - int(user_base)).astype(np.int64)
    db = items[:nb].cpu().numpy().astype(np.int64)
    pm2 = pm.reshape(pm.shape[0], -1)
    calls, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < min(budget_s, 10.0) * 0.5:
        z = pm2[ub] * Dt[db]
        h2 = np.maximum(np.maximum(z @ W1 + b1, 0) @ W2 + b2, 0)
        (z.sum(axis=1) + (h2 @ w3 + b3)).sum()
        calls += 1
    rate = calls * nb / (time.perf_counter() - t0)
    got = gpu_sample.cpu().numpy().astype(np.float64)
    err = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
    ok = bool(err <= PARITY_TOL and np.array_equal(np.isnan(got), np.isnan(ref)))
    return ({"value": rate, "unit": "pairs/s", "cores": usable_cores(), "kind": "port",
             "sample": "numpy float32 restatement of the build-defined head (gather, multiply, two BLAS GEMMs, dot) on %d-pair "
                       "slices of the timed batch, dish vectors built once, %d calls; parity: float64 restatement on the "
                       "first %d pairs" % (nb, calls, n),
             "max_rel_diff_vs_gpu": err, "parity_tolerance": PARITY_TOL, "parity_ok": ok}, ok)


def cpu_baseline(torch, PM, RE, CE, users, items, cats, budget_s):
    """CPU restatement of the reference graph (oracle/torch_graph.py) on this box's host cores.

    Times three call sizes of the same workload -- the reference's own 51 pairs per call
    (evaluate.py:39-58), 4096 and 65536 -- and reports the fastest as `value`, so the baseline is
    the most favourable batching of the op-for-op graph, not a strawman."""
    from oracle import c_oracle, torch_graph
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    Bc = min(1 << 18, users.numel())
    pm, re, ce = PM.cpu(), RE.cpu(), CE.cpu()
    u, d, m = users[:Bc].cpu(), items[:Bc].cpu(), cats[:Bc].cpu()
    ref = torch_graph.inference(pm, re, ce, u, d, m)                         # also the parity sample
    rates = {}
    for size in (51, 4096, 65536):
        calls, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s * 0.22:
            o = (calls * size) % (Bc - size)
            torch_graph.inference(pm, re, ce, u[o:o + size], d[o:o + size], m[o:o + size])
            calls += 1
        rates[size] = calls * size / (time.perf_counter() - t0)
    best = max(rates, key=rates.get)
    # context: the fused scalar C port of the same formula (no temporaries), all OpenMP threads
    pmn, ren, cen = pm.numpy(), re.numpy(), ce.numpy()
    un, dn, mn = u.numpy(), d.numpy(), m.numpy()
    cthreads = min(ncores, c_oracle.max_threads())
    c_oracle.score_pairs(pmn, ren, cen, un[:4096], dn[:4096], mn[:4096], nthreads=cthreads)
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s * 0.2:
        c_oracle.score_pairs(pmn, ren, cen, un, dn, mn, nthreads=cthreads)
        reps += 1
    c_rate = reps * Bc / (time.perf_counter() - t0)
    return {"value": rates[best], "unit": "pairs/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": "CPU restatement of reference graph (TF unavailable): torch-CPU op-for-op with [B,C,E] "
                      "temporaries on %d-pair slices of the same workload, ~%.0f s per call size; pairs/s at "
                      "51 / 4096 / 65536 pairs per call = %.3g / %.3g / %.3g (value = best, %d per call)"
                      % (Bc, budget_s * 0.22, rates[51], rates[4096], rates[65536], best),
            "value_51_pair_calls": rates[51],
            "host_cpu_count": os.cpu_count(),
            "fused_c_port": {"value": c_rate, "unit": "pairs/s", "cores": cthreads,
                             "what": "oracle/m2d_oracle.c, fused scalar loop, OpenMP"}}, ref, Bc


def catalogue_topk_leg(torch, eng, U, I, C, E, dev, user_base, n_users, k=10, keep=None):
    """Outside the timed region: full-catalogue top-k (m2d_topk_users, fp32 MFMA) for n_users users.  `keep`: a dict that
    receives the last call's lists (`compare_lists`)."""
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    pat = torch.randint(1, 2 ** C, (I,), generator=g, device=dev, dtype=torch.int32)
    dish_cats = ((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).to(torch.float32)
    eng.set_dish_categories(dish_cats)
    users = (torch.randperm(U, generator=g, device=dev)[:n_users].to(torch.int32) + int(user_base)).contiguous()
    eng.topk_users(users[:1024], k)                       # builds the retrieval tables
    t_warm = time.perf_counter()                          # the first full launches run 5-10 % slow (clock ramp): at least three,
    for i in range(40):                                   # and 60 ms of them (the every-tile form settled only after ~15 launches)
        eng.topk_users(users, k)
        if i >= 2:
            torch.cuda.synchronize()
            if time.perf_counter() - t_warm > 0.06:
                break
    torch.cuda.synchronize()
    reps = 7
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for i in range(reps):
        evs[i].record()
        last = eng.topk_users(users, k)
    evs[reps].record()
    torch.cuda.synchronize()
    eng.check()
    if keep is not None:
        keep["scores"], keep["ids"] = last
    ms = median([evs[i].elapsed_time(evs[i + 1]) for i in range(reps)])
    kernel = eng.last_kernel()
    dense = 2.0 * (C + 1) * E * n_users * I                 # the [users x (C+1)E] . [(C+1)E x dishes] contraction
    # the pattern-grouped kernel (0/1 masks) contracts over E only: price it on the flops it executes
    flops = 2.0 * E * n_users * I if kernel.startswith("m2d_topk_grouped") else dense
    x3 = kernel.endswith("bf16x3")                         # 3 bf16 MFMAs per 16 k-values: 6*E flop per pair on the bf16 pipe
    scanned = full = None
    if kernel.startswith("m2d_topk_grouped"):              # these kernels step through their blocks' relevant patterns only
        scanned, full = eng.get_option("topk_tiles_scanned"), eng.get_option("topk_tiles_full")
        if scanned > 0:
            flops = 2.0 * E * eng.get_option("topk_block_users") * 32 * scanned     # a block's user lanes (256, or 128) x a tile's 32 dishes
    # split bf16: three products per tile -- or, the hi x hi first form (large catalogues), one product per tile
    # and the two cross products for the (wave, tile) pairs that could still hold a candidate
    completed = eng.get_option("topk_tiles_completed") if kernel.startswith("m2d_topk_grouped") else -1
    ex3 = 3 * flops if completed < 0 else flops + 2 * (2.0 * E * 32 * 32 * completed)
    return {"users": n_users, "dishes": I, "k": k, "median_ms": ms, "users_per_s": n_users / ms * 1e3,
            "pairs_per_s": n_users * I / ms * 1e3, "tflops": flops / ms / 1e9,
            "dense_equivalent_tflops": dense / ms / 1e9,
            "roofline": ({"bound": "mfma", "achieved": ex3 / ms / 1e9, "peak": 2500.0, "unit": "TFLOP/s",
                          "frac": ex3 / ms / 1e9 / 2500.0, "flop_per_pair": ex3 / n_users / I,
                          "hi_first_form": completed >= 0, "wave_tiles_given_cross_products": (completed if completed >= 0 else None),
                          "frac_if_priced_as_three_products": (3 * flops / ms / 1e9 / 2500.0 if completed >= 0 else None),
                          "tiles_scanned": scanned, "tiles_without_pruning": full,
                          "scanned_fraction": (scanned / full if scanned and full else None),
                          "frac_if_every_tile_were_scanned": 3 * 2.0 * E * n_users * I / ms / 1e9 / 2500.0,
                          "dtype": ("split bf16 (x = hi + lo): hi x hi for every tile, lo x hi + hi x lo for the tiles that can hold a candidate (v_mfma_f32_32x32x16_bf16, fp32 accumulate)" if completed >= 0 else "split bf16 (x = hi + lo, 3 x v_mfma_f32_32x32x16_bf16, fp32 accumulate)"),
                          **bare_loop_fields(ex3 / ms / 1e9),
                          "note": "pipelined kernel; `frac` prices the flops EXECUTED: users are sorted by the mask patterns that "
                                  "can reach their top-k and a block steps through those patterns' tiles only (Cauchy-Schwarz bounds widened by the f32 / split-bf16 rounding of the sums of absolute terms; "
                                  "DESIGN.md 4.4), so most (user, dish) pairs are decided without being multiplied -- pairs_per_s "
                                  "counts every pair of the catalogue"} if x3 else
                         {"bound": "mfma", "achieved": flops / ms / 1e9, "peak": 157.3, "unit": "TFLOP/s",
                          "frac": flops / ms / 1e9 / 157.3, "dtype": "f32 (v_mfma_f32_32x32x2_f32, exact)",
                          "flop_per_pair": flops / n_users / I, "tiles_scanned": scanned, "tiles_without_pruning": full,
                          "scanned_fraction": (scanned / full if scanned and full else None),
                          "frac_if_every_tile_were_scanned": 2.0 * E * n_users * I / ms / 1e9 / 157.3,
                          "note": "`frac` prices the flops EXECUTED (tiles of the blocks' relevant patterns), as for the split-bf16 kernel"}),
            "kernel": kernel}


def compare_lists(torch, a, b):
    """Dish ids are index output: how the default (split-bf16) lists differ from the exact-f32 kernel's for the same users.
    Both kernels finish near-tied lists in one arithmetic (m2d_topk_refine; option topk_refine), so they should not; without
    it, where two dishes' scores sit inside the split's rounding the two kernels may order them differently; `max_gap_at_mismatch` is the largest |score difference| between the two kernels at a
    position that holds different dishes, relative to max(1, |score|)."""
    ia, ib, sa, sb = a["ids"], b["ids"], a["scores"], b["scores"]
    diff = ia != ib
    rows = diff.any(dim=1)
    gap = ((sa - sb).abs() / sb.abs().clamp(min=1.0))[diff]
    return {"lists_identical_frac": 1.0 - float(rows.float().mean().item()), "lists_differing": int(rows.sum().item()),
            "positions_differing": int(diff.sum().item()),
            "max_gap_at_mismatch": float(gap.max().item()) if gap.numel() else 0.0,
            "max_score_difference": float(((sa - sb).abs() / sb.abs().clamp(min=1.0)).nan_to_num(nan=0.0).max().item()),
            "what": "default split-bf16 lists against the exact-f32 kernel's (option topk_bf16x3 = 0), same users and dishes"}


class _Clock:
    """HIP events on the current stream for a GPU device, perf_counter on CPU (the gloo test of these legs)."""

    def __init__(self, torch, dev):
        self.torch, self.gpu = torch, torch.device(dev).type == "cuda"

    def mark(self):
        if self.gpu:
            e = self.torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        return time.perf_counter()

    def sync(self):
        if self.gpu:
            self.torch.cuda.synchronize()

    def ms(self, a, b):
        return a.elapsed_time(b) if self.gpu else (b - a) * 1e3


def sharded_topk_leg(torch, dist, eng, U, I, C, E, dev, user_base, n_users, world, k=10, repeats=7):
    """Every rank: top-k over the replicated catalogue for n_users of ITS users, then ONE all-gather of
    [n_users, k] x (f32 score, i32 id) per rank, through foodrec_amd.sharding.  Timed
    `repeats` times between barriers; the median of the max-over-ranks wall time is reported.  `dist` is None in a
    single-process run (N = 1 without a launcher): the same leg with no peers, so that the N = 1 line carries the
    number the N > 1 lines are compared with."""
    from foodrec_amd.sharding import UserShardedScorer
    g = torch.Generator(device=dev)
    g.manual_seed(11)                                     # same dish masks on every rank (replicated)
    pat = torch.randint(1, 2 ** C, (I,), generator=g, device=dev, dtype=torch.int32)
    dish_cats = ((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).to(torch.float32)
    eng.set_dish_categories(dish_cats)
    sh = UserShardedScorer(eng, world * U, device=dev, always_collective=dist is not None)
    users = (torch.randperm(U, generator=g, device=dev)[:n_users].to(torch.int32) + int(user_base)).contiguous()
    sh.topk_users_gathered(users[:1024], k)               # builds the retrieval tables, warms RCCL up
    for _ in range(5):                                    # the first full launches run 5-10 % slow (clock ramp)
        sh.topk_users_gathered(users, k)
    walls, tk_ms, ag_ms = [], [], []
    clk = _Clock(torch, dev)
    for _ in range(repeats):
        clk.sync()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        e0 = clk.mark()
        s, ids = sh.topk_local(k, users)
        e1 = clk.mark()
        gs, gi = sh._gather_topk(s, ids, n_users, k) if dist is not None else (s, ids)
        e2 = clk.mark()
        clk.sync()
        wall = time.perf_counter() - t0
        t = torch.tensor([wall, clk.ms(e0, e1), clk.ms(e1, e2)], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        w, a_, b_ = (float(x) for x in t.tolist())
        walls.append(w); tk_ms.append(a_); ag_ms.append(b_)
    eng.check()
    r = dist.get_rank() if dist is not None else 0
    ok = bool(torch.equal(gi[r * n_users:(r + 1) * n_users], ids) and torch.equal(gs[r * n_users:(r + 1) * n_users], s))
    wall = median(walls)
    return {"users_per_gpu": n_users, "dishes": I, "k": k, "repeats": repeats, "wall_ms_median": wall * 1e3,
            "wall_ms_all": [w * 1e3 for w in walls], "topk_ms_median": median(tk_ms), "allgather_ms_median": median(ag_ms),
            "allgather_bytes_per_rank": n_users * k * 8 if dist is not None else 0,
            "users_per_s_whole_job": world * n_users / wall, "pairs_per_s_whole_job": world * n_users * I / wall,
            "kernel": eng.last_kernel(), "own_slice_roundtrip_ok": ok,
            "what": "max over ranks per repeat, median over repeats; per-shard full-catalogue top-k + one RCCL all-gather"
                    + ("" if dist is not None else " (single process: no peers, no collective)")}


def world_rows_ok(torch, sh, gs, gi):
    """The gathered result's padding: rows of rank r beyond its shard's count do not exist (the result is trimmed to the users
    that do), and every existing row holds k distinct dish ids >= 0 -- a cheap check of the OTHER ranks' slices (their content
    is checked by the rank that owns them)."""
    if gi.shape[0] != sh.num_users_total:
        return False
    return bool((gi >= 0).all())


def sharded_all_users_leg(torch, dist, sh, I, k, round_users, repeats=1, warm_rounds=2):
    """The user-sharded top-k path as north_star states it: every rank ranks EVERY user of its shard over the replicated
    catalogue in rounds of `round_users` users and the ranks exchange their final lists -- [shard, k] x (f32 score, i32 id)
    per rank -- by all-gather, one piece per round, each issued asynchronously while the next round is being ranked
    (foodrec_amd.sharding.UserShardedScorer.topk_all_users): only the last round's exchange is exposed.  `sh` is a
    UserShardedScorer; `dist` is None in a single-process run (no peers, no collective).  Wall time = max over ranks,
    median over repeats; `allgather_exposed_ms` = what the stream still waited for after the last round's kernels."""
    clk = _Clock(torch, sh.device)
    per_round = min(int(round_users), max(sh.count, 1))
    first = torch.arange(sh.base, sh.base + min(per_round, sh.count), dtype=torch.int32, device=sh.device)
    if sh.count:
        sh.topk_local(k, first)                            # builds the retrieval tables
        for _ in range(warm_rounds):
            sh.topk_local(k, first)
    if dist is not None:                                  # the collective's buffers and connections, once, at their real sizes
        sh.topk_all_users(k, round_users=round_users)
    walls, exposed = [], []
    ok = True
    for _ in range(repeats):
        clk.sync()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        if dist is not None:
            gs, gi = sh.topk_all_users(k, round_users=round_users)
        else:
            gs, gi = sh.topk_local_rounds(k, round_users)
        clk.sync()
        ex = 0.0
        if dist is not None and getattr(sh, "last_allgather_events", None):
            ex = sh.last_allgather_events[0].elapsed_time(sh.last_allgather_events[1])
        t = torch.tensor([time.perf_counter() - t0, ex], dtype=torch.float64, device=sh.device)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        w, e_ = (float(x) for x in t.tolist())
        walls.append(w); exposed.append(e_)
    if sh.count:
        # Outside the timed region: this rank's WHOLE shard ranked again on its own, without any collective, must sit in the
        # gathered result bit for bit -- every round, the buffer-reusing ones (index >= 2) and a short last one included.  (Round 4
        # looked at round 0 only, which never reuses a staging buffer: a stream-ordering fault of the pipelined gather on real
        # RCCL would have passed.)
        ls, li = sh.topk_local_rounds(k, round_users)
        lo = sh.rank * sh.per if dist is not None else 0
        ok = bool(torch.equal(gi[lo:lo + sh.count], li) and
                  torch.equal(gs[lo:lo + sh.count].view(torch.int32), ls.view(torch.int32)))
        if dist is not None and world_rows_ok(torch, sh, gs, gi) is False:
            ok = False
    if sh.scorer is not None:
        sh.scorer.check()
    wall = median(walls)
    total_users = sh.num_users_total
    return {"path": "sharded_topk_allgather", "users_total": total_users, "users_per_gpu": sh.per, "dishes": I, "k": k,
            "round_users": int(round_users), "rounds_per_gpu": -(-sh.per // int(round_users)), "repeats": repeats,
            "wall_ms": wall * 1e3, "allgather_exposed_ms": median(exposed) if dist is not None else 0.0,
            "allgather": ("one asynchronous all-gather per round, overlapped with the next round's ranking; exposed = the last "
                          "round's exchange and its copy into the result") if dist is not None else "none (single process)",
            "allgather_bytes_per_rank": sh.per * k * 8 if dist is not None else 0,
            "users_per_s_whole_job": total_users / wall, "pairs_per_s_whole_job": total_users * I / wall,
            "own_slice_roundtrip_ok": ok}


def scaling_path_block(torch, dist, foodrec_amd, dev, world, rank, users_total, I, E, k, round_users, repeats=1):
    """`scaling_path`: the sharded top-k path at BASELINE configs[3] / configs[4]'s per-GPU shape, on tables of its own
    (users_total / world users per GPU x I replicated dishes).  The split-bf16 kernel (the default) over every user of
    the shard + the all-gather; the exact-f32 kernel's rate beside it, measured on one round of users per GPU."""
    from foodrec_amd.sharding import UserShardedScorer, shard_range
    C = 4
    base, count = shard_range(users_total, world, rank)
    g = torch.Generator(device=dev); g.manual_seed(20260101 + 4)             # replicated tables: the same on every rank
    sc = E ** -0.5
    RE = torch.randn((I, E), generator=g, device=dev) * sc
    CE = torch.randn((C, E), generator=g, device=dev) * sc
    pat = torch.randint(1, 2 ** C, (I,), generator=g, device=dev, dtype=torch.int32)
    dish_cats = ((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).float()
    g.manual_seed(20260101 + 40 + rank)
    PM = torch.randn((max(count, 1), C + 1, E), generator=g, device=dev) * sc
    eng = foodrec_amd.ScoringEngine(PM, RE, CE, coef=0.99, device=dev, user_base=base)
    eng.set_dish_categories(dish_cats)
    sh = UserShardedScorer(eng, users_total, device=dev, always_collective=dist is not None)
    out = sharded_all_users_leg(torch, dist, sh, I, k, round_users, repeats=repeats)
    out["kernel"] = eng.last_kernel()
    x3 = out["kernel"].endswith("bf16x3")
    out["dtype"] = "bf16x3 (x = hi + lo, 3 x v_mfma_f32_32x32x16_bf16, fp32 accumulate)" if x3 else "f32"
    out["embed_size"] = E
    flop = 2.0 * E * (3 if x3 else 1)                      # per (user, dish) on the pattern-grouped kernels
    out["roofline_frac_of_mfma_peak"] = flop * out["pairs_per_s_whole_job"] / world / 1e12 / (2500.0 if x3 else 157.3)
    out["repaired_users_last_round"] = eng.get_option("topk_repaired")
    if x3:
        sc_, fl_ = eng.get_option("topk_tiles_scanned"), eng.get_option("topk_tiles_full")
        out["scanned_fraction_last_round"] = sc_ / fl_ if fl_ else None
        out["roofline_frac_of_mfma_peak"] = (out["roofline_frac_of_mfma_peak"] * sc_ / fl_) if fl_ else out["roofline_frac_of_mfma_peak"]
        out["roofline_note"] = ("fraction of the dense bf16 MFMA peak on the flops executed (tiles stepped through x 3 MFMAs); "
                                "pairs_per_s_whole_job counts every (user, dish) pair of the catalogue")
    # pairs DECIDED (every pair of the catalogue: most by a bound, without being multiplied) and pairs MULTIPLIED (the tiles
    # the blocks stepped through; the last round's share stands for the shard)
    out["pairs_decided_per_s_whole_job"] = out["pairs_per_s_whole_job"]
    out["pairs_multiplied_per_s_whole_job"] = out["pairs_per_s_whole_job"] * (out.get("scanned_fraction_last_round") or 1.0)
    # the exact-f32 kernel on one round of this shard's users (every rank at once; max over ranks)
    clk = _Clock(torch, dev)
    eng.set_option("topk_bf16x3", 0)
    n1 = min(int(round_users), count)
    users = torch.arange(base, base + n1, dtype=torch.int32, device=dev)
    ms = []
    if n1:
        eng.topk_users(users, k)
        for _ in range(3):
            clk.sync()
            if dist is not None:
                dist.barrier()
            a = clk.mark(); eng.topk_users(users, k); b = clk.mark(); clk.sync()
            t = torch.tensor([clk.ms(a, b)], dtype=torch.float64, device=dev)
            if dist is not None:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms.append(float(t.item()))
        eng.check()
    if ms:
        m = median(ms)
        sc_, fl_ = eng.get_option("topk_tiles_scanned"), eng.get_option("topk_tiles_full")
        part = sc_ / fl_ if (fl_ and sc_) else 1.0           # tiles stepped through / all tiles (pattern pruning)
        out["exact_f32"] = {"kernel": eng.last_kernel(), "users_per_gpu_in_sample": n1, "topk_ms": m,
                            "pairs_per_s_whole_job": world * n1 * I / m * 1e3,
                            "roofline_frac_of_f32_mfma_peak": part * 2.0 * E * n1 * I / m / 1e9 / 157.3,
                            "scanned_fraction": part,
                            "what": "option topk_bf16x3 = 0 (v_mfma_f32_32x32x2_f32, exact): one round of users per GPU, all "
                                    "ranks at once, no all-gather; whole-shard time = this rate x the shard"}
    out["what"] = ("BASELINE configs[%d] per-GPU shape: %d users over %d GPU(s) x %d replicated dishes, E = %d; per-shard "
                   "top-%d for every user in rounds of %d + ONE all-gather of [shard, %d] x (f32, i32)%s; max over ranks"
                   % (3 if E == 64 else 4, users_total, world, I, E, k, round_users, k,
                      "" if dist is not None else " (single process: no peers, no collective)"))
    eng.close()
    del PM, RE, CE, dish_cats, eng, sh
    if torch.device(dev).type == "cuda":
        torch.cuda.empty_cache()
    return out


XGMI_LINK_GBS = 153.0      # one xGMI link, per direction (7 links per GPU, point to point)
# What the matrix pipe sustains on this part in a loop of nothing but v_mfma_f32_32x32x16_bf16 from registers, every CU, two waves
# per SIMD (scripts/diag/mfma_chain_probe.cpp, profiles/r05_mfma_chain_probe.txt): the clock it holds depends on the operands.
BARE_BF16_MFMA_LOOP = {"zero_operands_TFLOPs": 2460.0, "random_operands_TFLOPs": 1865.0,
                       "source": "profiles/r05_mfma_chain_probe.txt: 2.38 GHz on zeros, 1.83 GHz on N(0, 1) bf16 operands; `peak` stays "
                                 "the spec figure (2 500 at 2.4 GHz), these say how much of it a power-limited part can be asked for"}


def bare_loop_fields(achieved_tflops):
    return {"bare_mfma_loop": BARE_BF16_MFMA_LOOP,
            "frac_of_bare_mfma_loop_random_operands": achieved_tflops / BARE_BF16_MFMA_LOOP["random_operands_TFLOPs"]}




def projected_world8_block(torch, foodrec_amd, dev, users_total, I, E, k, topk_path_ms_n1, rounds=(262144, 0, 524288), repeats=3):
    """A ONE-GPU PROJECTION of the sharded top-k path at 8 GPUs -- not a measurement of 8 GPUs: this box has one.  What one
    GPU can say: how long the N = 8 per-GPU shape takes (BASELINE configs[3]: users_total / 8 users held as the LAST shard of
    eight, the same replicated catalogue, ranked in rounds), at several round sizes -- the fixed launches of a retrieval call and
    a short last round weigh more on a shard an eighth the size.  What it cannot say is what the seven peers and the
    collective do; the exchange is priced from SURVEY.md section 5's link rate instead.  `rounds`: users per round; 0 = the shard
    cut into the fewest EVEN rounds of at most 524 288."""
    from foodrec_amd.sharding import UserShardedScorer, shard_range
    C, world, rank = 4, 8, 7
    base, count = shard_range(users_total, world, rank)
    per = -(-users_total // world)
    g = torch.Generator(device=dev); g.manual_seed(20260101 + 4)             # the replicated tables of scaling_path_block
    sc = E ** -0.5
    RE = torch.randn((I, E), generator=g, device=dev) * sc
    CE = torch.randn((C, E), generator=g, device=dev) * sc
    pat = torch.randint(1, 2 ** C, (I,), generator=g, device=dev, dtype=torch.int32)
    dish_cats = ((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).float()
    g.manual_seed(20260101 + 40 + rank)
    PM = torch.randn((count, C + 1, E), generator=g, device=dev) * sc
    eng = foodrec_amd.ScoringEngine(PM, RE, CE, coef=0.99, device=dev, user_base=base)
    eng.set_dish_categories(dish_cats)
    sh = UserShardedScorer(eng, users_total, device=dev)
    sh.base, sh.count, sh.per, sh.rank, sh.world = base, count, per, rank, world      # this process plays rank 7 of 8 (no collective is issued)
    clk = _Clock(torch, dev)
    out_rounds = []
    for R in rounds:
        R = int(R) if R else -(-count // -(-count // 524288))
        first = torch.arange(base, base + min(R, count), dtype=torch.int32, device=dev)
        sh.topk_local(k, first); sh.topk_local(k, first)                      # tables, scratch at this round's size
        walls = []
        for _ in range(repeats):
            clk.sync()
            t0 = time.perf_counter()
            sh.topk_local_rounds(k, R)
            clk.sync()
            walls.append((time.perf_counter() - t0) * 1e3)
        eng.check()
        nround = -(-count // R)
        last_rows = count - (nround - 1) * R
        piece, last_piece = R * k * 8, last_rows * k * 8
        shard_ms = median(walls)
        # all-gather of one round's pieces over xGMI: every rank sends its piece to 7 peers over 7 links at once (direct, what a
        # fully connected topology allows) or around a ring (7 steps of one piece per link)
        direct_ms, ring_ms = piece / (XGMI_LINK_GBS * 1e6), 7 * piece / (XGMI_LINK_GBS * 1e6)
        exposed_direct, exposed_ring = last_piece / (XGMI_LINK_GBS * 1e6), 7 * last_piece / (XGMI_LINK_GBS * 1e6)
        out_rounds.append({
            "round_users": R, "rounds": nround, "last_round_users": last_rows, "shard_ms": shard_ms, "shard_ms_all": walls,
            "ms_per_round_avg": shard_ms / nround, "allgather_bytes_per_rank_per_round": piece,
            "allgather_ms_per_round_at_153GBps_direct": direct_ms, "allgather_ms_per_round_at_153GBps_ring": ring_ms,
            "hidden_behind_next_round": bool(ring_ms < shard_ms / nround),
            "exposed_last_round_ms_direct": exposed_direct, "exposed_last_round_ms_ring": exposed_ring,
            "implied_speedup_upper_bound": (topk_path_ms_n1 / (shard_ms + exposed_direct)) if topk_path_ms_n1 else None,
            "implied_speedup_with_ring_exchange": (topk_path_ms_n1 / (shard_ms + exposed_ring)) if topk_path_ms_n1 else None})
    best = min(out_rounds, key=lambda r: r["shard_ms"])
    eng.close()
    del PM, RE, CE, dish_cats, eng, sh
    if torch.device(dev).type == "cuda":
        torch.cuda.empty_cache()
    return {"status": "PROJECTION from one GPU: UNMEASURED ON HARDWARE at N = 8",
            "what": ("the N = 8 per-GPU shape of the sharded top-k path timed on ONE GPU: %d of %d users held as shard [%d, %d), "
                     "%d replicated dishes, E = %d, top-%d for every user of the shard in rounds; the exchange priced at %.0f GB/s "
                     "per xGMI link (SURVEY.md section 5), not run" % (count, users_total, base, base + count, I, E, k, XGMI_LINK_GBS)),
            "topk_path_ms_n1": topk_path_ms_n1, "users_per_gpu": count, "rounds": out_rounds,
            "best_round_users": best["round_users"], "shard_ms": best["shard_ms"],
            "implied_speedup_upper_bound": best["implied_speedup_upper_bound"],
            "implied_speedup_with_ring_exchange": best["implied_speedup_with_ring_exchange"],
            "upper_bound_because": ("every rank is assumed as fast as this one, the per-round collectives fully hidden behind the next "
                                    "round's ranking (they take a few per cent of a round at the link rate), launch and host overheads "
                                    "as on this box; north_star asks for >= 6x"),
            "north_star_target": 6.0}


def default_round_users(per_gpu_users, requested):
    """--round-users 0 (the default): the shard cut into the fewest EVEN rounds of at most 524 288 users -- one round of 524 288
    at N = 1's 10 M users (20 of them), three of 416 667 at N = 8's 1.25 M (a short last round pays a retrieval call's fixed
    launches for a fraction of the work: scaling_path.projected_world8 measures the difference)."""
    if requested:
        return int(requested)
    if per_gpu_users <= 0:
        return 524288
    return -(-per_gpu_users // -(-per_gpu_users // 524288))


def routed_pairs_leg(torch, dist, eng, U, I, C, dev, world, B, repeats=5):
    """Every rank brings B pairs whose users are spread over ALL shards; UserShardedScorer.score_pairs_routed buckets
    them by owner, all-to-alls the records, the owners score, the scores come back ('pairs routed to the
    owner of the user').  Whole-job pairs/s over the median max-over-ranks wall time."""
    from foodrec_amd.sharding import UserShardedScorer
    sh = UserShardedScorer(eng, world * U, device=dev, always_collective=True)
    g = torch.Generator(device=dev)
    g.manual_seed(900 + dist.get_rank())
    users = torch.randint(0, world * U, (B,), generator=g, device=dev, dtype=torch.int32)
    items = torch.randint(0, I, (B,), generator=g, device=dev, dtype=torch.int32)
    pat = torch.randint(1, 2 ** C, (B,), generator=g, device=dev, dtype=torch.int32)
    cats = ((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).to(torch.float32).contiguous()
    sh.score_pairs_routed(users, items, cats)             # warm-up, with the collective id check
    walls = []
    clk = _Clock(torch, dev)
    for _ in range(repeats):
        clk.sync()
        dist.barrier()
        t0 = time.perf_counter()
        out = sh.score_pairs_routed(users, items, cats, check=False)
        clk.sync()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        walls.append(float(t.item()))
    sh.check()
    idx = (sh.owner_of(users) == dist.get_rank()).nonzero(as_tuple=True)[0]      # the pairs this rank owns: same kernel, same bits
    ok = bool(idx.numel() == 0 or torch.equal(eng.score_pairs(users[idx], items[idx], cats[idx]), out[idx]))
    eng.check()
    wall = median(walls)
    return {"pairs_per_gpu": B, "repeats": repeats, "wall_ms_median": wall * 1e3, "wall_ms_all": [w * 1e3 for w in walls],
            "pairs_per_s_whole_job": world * B / wall, "bytes_per_pair_on_the_wire": (2 + C) * 4 + 4,
            "own_pairs_match_local_scoring": ok,
            "what": "bucket by owner (one device sort) + all-to-all of (user, dish, mask) records + owner-side "
                    "m2d_score_pairs + all-to-all of f32 scores; includes the one host round trip for bucket sizes"}


def evaluator_leg(torch, dev, budget_s=6.0):
    """BASELINE configs[0] shape (U = 64 657, I = 4 548, C = 4, E = 32; Train_recommender.py:51-60): the
    batched device evaluator (one m2d_rank_candidates launch for all users) beside the reference's loop
    structure -- one scoring call of 51 pairs + heapq per user (evaluate.py:28-66) -- run on the CPU
    restatement for a sample of users."""
    import types
    import numpy as np
    import foodrec_amd
    from foodrec_amd import formats
    from oracle import m2d_oracle, torch_graph
    U, I, C, E, K = 64657, 4548, 4, 32, 10
    pm, re, ce, _, cats = formats.synthetic_tables(U, I, C, E, 95, seed=20260101 + 1)
    rng = np.random.default_rng(5)
    pos = rng.integers(0, I, U)
    neg = rng.integers(0, I, (U, 100))
    ratings = {str(u): [int(pos[u])] for u in range(U)}
    negatives = {str(u): neg[u].tolist() for u in range(U)}
    d2c = {str(d): [[float(x)] for x in cats[d]] for d in range(I)}
    args = types.SimpleNamespace(num_categories=C, num_users=U, embed_size=E, high_level_score_coefficient=0.99)
    model = foodrec_amd.Model(args, pm, re, ce, None, device=dev)
    foodrec_amd.evaluate_model(None, model, {k: ratings[k] for k in list(ratings)[:64]}, negatives, K, d2c)   # warm
    foodrec_amd.clear_eval_plans()
    t0 = time.perf_counter()
    hits, ndcgs = foodrec_amd.evaluate_model(None, model, ratings, negatives, K, d2c)      # builds the device plan
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    hits2, ndcgs2 = foodrec_amd.evaluate_model(None, model, ratings, negatives, K, d2c)    # every later epoch: plan reused
    t_dev2 = time.perf_counter() - t0
    # device part alone (ids already on the device): one launch
    users_t = torch.arange(U, dtype=torch.int32, device=dev)
    items_t = torch.from_numpy(np.concatenate([pos[:, None], neg[:, 50:100]], axis=1).astype(np.int32)).to(dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    model.engine.rank_candidates(users_t, items_t, K)
    e1.record()
    torch.cuda.synchronize()
    pmt, ret, cet = torch.from_numpy(pm), torch.from_numpy(re), torch.from_numpy(ce)
    fn = lambda u, i, c: torch_graph.inference(pmt, ret, cet, torch.tensor([int(x) for x in u]), torch.tensor(i),
                                               torch.tensor(c, dtype=torch.float32)).numpy()
    n, t1 = 0, time.perf_counter()
    keys = list(ratings)
    rh = []
    while time.perf_counter() - t1 < budget_s and n < U:
        sub = {k: ratings[k] for k in keys[n:n + 200]}
        h, _ = m2d_oracle.evaluate_model(fn, sub, negatives, K, d2c)
        rh += h
        n += 200
    t_cpu = time.perf_counter() - t1
    return {"users": U, "dishes": I, "embed_size": E, "candidates_per_user": 51, "K": K,
            "device_evaluate_model_s": t_dev, "device_users_per_s": U / t_dev,
            "device_evaluate_model_second_call_s": t_dev2, "device_users_per_s_second_call": U / t_dev2,
            "second_call_identical": bool(hits2 == hits and ndcgs2 == ndcgs),
            "device_rank_launch_ms": e0.elapsed_time(e1), "device_pairs_per_s_in_launch": U * 51 / e0.elapsed_time(e1) * 1e3,
            "cpu_reference_loop_users_per_s": n / t_cpu, "cpu_sample_users": n,
            "hr_at_10": float(np.mean(hits)), "ndcg_at_10": float(np.mean(ndcgs)),
            "hr_matches_cpu_on_sample": bool(hits[:len(rh)] == rh),
            "what": "evaluate.py:13-66 on synthetic files of the reference's default sizes; device first call = host list "
                    "building + H2D + one m2d_rank_candidates launch; second call = the cached device plan (what every "
                    "later epoch costs, Train_recommender.py:210); cpu = one 51-pair scoring call + heapq per user on the "
                    "CPU restatement"}


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def side_measurements(torch, eng, PM, U, I, C, E, dev, user_base):
    """Outside the timed region: (i) the same kernel on a batch in which every user occurs at most once
    (no cache reuse of Personal_Memory rows at all), (ii) a plain streaming read of Personal_Memory."""
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    Bn = min(U, 1 << 20)
    users = (torch.randperm(U, generator=g, device=dev)[:Bn].to(torch.int32) + int(user_base)).contiguous()
    items = torch.randint(0, I, (Bn,), generator=g, device=dev, dtype=torch.int32)
    cats = torch.ones((Bn, C), device=dev)
    out = torch.empty(Bn, device=dev)
    time_steps(torch, eng, users, items, cats, out, 3)
    _, per = time_steps(torch, eng, users, items, cats, out, 20)
    ms = median(per)
    bpp = algorithmic_bytes_per_pair(C, E)
    nr = {"pairs_per_launch": Bn, "kernel_median_ms": ms, "achieved": bpp * Bn / ms / 1e6, "unit": "GB/s",
          "frac": bpp * Bn / ms / 1e6 / HBM_PEAK_GBS,
          "what": "same kernel, every user at most once per launch (randperm) -> no Personal_Memory reuse"}
    sink = torch.zeros(4, device=dev)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(12)]
    nbytes = PM.numel() * 4
    for i in range(11):
        evs[i].record()
        eng.stream_read_probe(PM, sink)
    evs[11].record()
    torch.cuda.synchronize()
    ms = median([evs[i].elapsed_time(evs[i + 1]) for i in range(1, 11)])
    probe = {"bytes": nbytes, "median_ms": ms, "GBps": nbytes / ms / 1e6,
             "what": "m2d_stream_read_probe: plain 16 B/lane streaming read of Personal_Memory"}
    # HBM-only estimate from the no-reuse leg: what cannot come from the Infinity Cache is the Personal_Memory blocks
    # (each read once per launch from a table far larger than the cache) plus the id / mask / score streams; the dish
    # rows (a table of %d MB) are re-read on-die and are left out
    pm_bytes, stream_bytes = Bn * (C + 1) * E * 4, Bn * (C * 4 + 12)
    t = nr["kernel_median_ms"]
    re_cached = I * E * 4 <= INFINITY_CACHE_BYTES // 2
    hb = (pm_bytes + stream_bytes + (0 if re_cached else Bn * E * 4)) / t / 1e6
    hbm_only = {"achieved": hb, "unit": "GB/s", "frac_of_spec_peak": hb / HBM_PEAK_GBS, "frac_of_stream_probe": hb / probe["GBps"],
                "bytes_per_launch": pm_bytes + stream_bytes + (0 if re_cached else Bn * E * 4), "kernel_median_ms": t,
                "what": "no-reuse leg, bytes that must come from HBM only: Personal_Memory blocks + id/mask/score streams%s"
                        % (" (dish rows excluded: the %.0f MB dish table is Infinity-Cache resident)" % (I * E * 4 / 1e6)
                           if re_cached else " + dish rows (the dish table does not fit the Infinity Cache)")}
    # the same with the benchmark's masks (random non-empty subsets): rows of absent categories are not fetched, so the
    # bytes that must come from HBM are U_high + the active rows
    g2 = torch.Generator(device=dev); g2.manual_seed(8)
    pat = torch.randint(1, 2 ** C, (Bn,), generator=g2, device=dev, dtype=torch.int32)
    cats2 = ((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).to(torch.float32).contiguous()
    skip = eng.get_option("skip_masked") != 0
    act = float(cats2.sum(1).mean().item()) if skip else float(C)
    time_steps(torch, eng, users, items, cats2, out, 3)
    _, per2 = time_steps(torch, eng, users, items, cats2, out, 20)
    t2 = median(per2)
    pm2 = Bn * (1.0 + act) * E * 4
    hb2 = (pm2 + stream_bytes + (0 if re_cached else Bn * E * 4)) / t2 / 1e6
    hbm_only["masked"] = {"achieved": hb2, "unit": "GB/s", "frac_of_spec_peak": hb2 / HBM_PEAK_GBS,
                          "frac_of_stream_probe": hb2 / probe["GBps"], "mean_active_categories": act,
                          "kernel_median_ms": t2, "pairs_per_s": Bn / t2 * 1e3,
                          "what": "the same no-reuse batch with the benchmark's masks (uniform non-empty subsets): HBM bytes = "
                                  "U_high + the rows of the active categories + streams"}
    return nr, probe, hbm_only


def user_high_leg(torch, eng, users, items, cats, C, E):
    """Outside the timed region: the same batch with the serving option "user_high_table" (the high-level sum from the
    derived table <U_high[u], CE_c>, 16 B per pair, instead of the gathered U_high row).  Not the headline: the table
    keeps part of the forward pass across launches."""
    out = torch.empty(users.numel(), dtype=torch.float32, device=users.device)
    eng.set_option("user_high_table", 1)
    try:
        time_steps(torch, eng, users, items, cats, out, 3)
        _, per = time_steps(torch, eng, users, items, cats, out, 20)
        eng.check()
        kern = eng.last_kernel()
    finally:
        eng.set_option("user_high_table", 0)
    ms = median(per)
    B = users.numel()
    active = float((cats != 0).sum(1).float().mean().item()) if eng.get_option("skip_masked") != 0 else float(C)
    bpp = (1.0 + active) * E * 4 + 2 * C * 4 + 12
    return {"kernel": kern, "kernel_median_ms": ms, "pairs_per_s": B / ms * 1e3, "algorithmic_bytes_per_pair": bpp,
            "achieved": bpp * B / ms / 1e6, "unit": "GB/s", "frac": bpp * B / ms / 1e6 / HBM_PEAK_GBS,
            "what": "option user_high_table = 1: sum_c m_c <U_high[u], CE_c> / n from a 16 B-per-user derived table instead "
                    "of the gathered E x 4-byte U_high row; same scores within 1e-6"}


def ingredients_leg(torch, eng, users, items, cats, I, C, E, dev, R):
    """Outside the timed region: the same batch with BASELINE configs[1]'s 10k-row ingredient table on the high-level
    path (build-defined extension; --workload ingredients makes it the timed step)."""
    g = torch.Generator(device=dev); g.manual_seed(20260101 + 3)
    lens = torch.randint(1, 21, (I,), generator=g, device=dev)
    off = torch.zeros(I + 1, dtype=torch.int32, device=dev)
    off[1:] = torch.cumsum(lens, 0).to(torch.int32)
    eng.set_ingredients(torch.randn((R, E), generator=g, device=dev) * E ** -0.5, off,
                        torch.randint(0, R, (int(off[-1].item()),), generator=g, device=dev, dtype=torch.int32))
    out = torch.empty(users.numel(), dtype=torch.float32, device=dev)      # its own buffer: never the timed kernel's
    step = lambda: eng.score_pairs_ingredients(users, items, cats, out=out)
    time_steps(torch, eng, users, items, cats, out, 3, step)
    _, per = time_steps(torch, eng, users, items, cats, out, 10, step)
    eng.check()
    kern = eng.last_kernel()
    eng.clear_ingredients()
    ms = median(per)
    B = users.numel()
    active = float((cats != 0).sum(1).float().mean().item()) if eng.get_option("skip_masked") != 0 else float(C)
    bpp = (3.0 + active) * E * 4 + C * 4 + 12          # U_high + H[d] + RE[d] + the low-level rows of the active categories
    return {"ingredient_rows": R, "ingredients_per_dish": "uniform 1..20", "kernel": kern, "kernel_median_ms": ms,
            "pairs_per_s": B / ms * 1e3, "algorithmic_bytes_per_pair": bpp, "achieved": bpp * B / ms / 1e6, "unit": "GB/s",
            "frac": bpp * B / ms / 1e6 / HBM_PEAK_GBS,
            "what": "same pairs, high-level path from the per-dish multi-hot ingredient sum H[d] (segment-sum hoisted to a "
                    "per-table kernel, DESIGN.md 8.1); no reference counterpart"}


def train_workload(a, torch, foodrec_amd, dev):
    """Single-GPU training-step throughput.  Default shape = the reference's flags
    (Train_recommender.py:35, :51-58): 64 657 users, 4 548 dishes, E = 200, batch 128."""
    import sys as _sys
    given = lambda name: any(x == name or x.startswith(name + "=") for x in _sys.argv)
    U = a.users if given("--users") else 64657
    I = a.dishes if given("--dishes") else 4548
    E = a.embed if given("--embed") else 200
    B = a.pairs if given("--pairs") else 128
    C = 4
    PM, RE, CE, users, items, cats = make_inputs(torch, dev, U, I, C, E, B, 20260101 + 6, 0)
    labels = (torch.rand(B, device=dev) < 0.5).float()
    eng = foodrec_amd.ScoringEngine(PM, RE, CE, coef=0.99, device=dev)
    eng.train_begin(a.learner, 0.001)
    step = lambda: eng.train_step(users, items, cats, labels)
    for _ in range(a.warmup):
        step()
    eng.check()
    wall, per = time_steps(torch, eng, users, items, cats, None, a.steps, step)
    eng.check()
    avg_ms = sum(per) / len(per)
    table_bytes = 4 * (PM.numel() + RE.numel() + CE.numel())
    dense = a.learner.lower() == "adam"
    # Adam (TF 1.x, not lazy): var, m, v of EVERY row read and written.  Others: the batch's rows only.
    pair_bytes = (2 * (C + 2) * E * 4 + C * 4 + 12) * B          # forward gather + gradient rows out
    alg = (6 * table_bytes if dense else 0) + pair_bytes
    ach = alg / (avg_ms * 1e-3) / 1e9
    line = {"metric": "trained (user,dish) pairs/sec", "value": B * a.steps / wall, "unit": "pairs/s", "n_gpus": 1,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "training step of Model_Recommender.py:99-104, :223-241 (sigmoid-CE loss, gradients, "
                                   "global-norm clip 5.0, %s update as TF 1.x applies it) on %d users x %d dishes, C=4, "
                                   "E=%d, batch %d; NOT the headline metric (SURVEY.md 8f row N4)" % (a.learner, U, I, E, B),
                       "users": U, "dishes": I, "embed_size": E, "batch": B, "learner": a.learner},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": None, "step_avg_ms": avg_ms, "algorithmic_bytes_per_step": alg,
                         "note": ("whole step (claim + grad + reduce + finalize + 3 apply + 2 cleanup launches) over the bytes the "
                                  "update rule must move: 6 x table bytes for TF 1.x Adam, which decays and moves every "
                                  "row every step" if dense else
                                  "whole step over the batch rows' bytes; launch-bound at this batch size")}}
    print(json.dumps(line))


def main():
    a = parse()
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # started by torch.distributed.run
    if a.gpus > 1 and not launched:
        sys.exit(launch_ranks(a))                                        # before torch / the GPU is touched
    if a.dry_run:
        return dry_run(a)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = launched
    # M2D_BENCH_REHEARSE_ONE_GPU=1: every rank on cuda:0, torch.distributed over gloo (device tensors) -- a FUNCTIONAL run of the
    # N > 1 legs on a one-GPU box (RCCL wants a GPU per rank); the line says so and its timings mean nothing
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

    if not os.path.exists(os.path.join(ROOT, "foodrec_amd", "libm2d.so")):      # clean checkout: hipcc, ~15 s
        if local == 0:
            import __graft_entry__
            __graft_entry__.build()
        if use_dist:
            dist.barrier()
    import foodrec_amd
    if a.workload == "train":
        if rank == 0:
            train_workload(a, torch, foodrec_amd, dev)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    if a.config is not None:                              # BASELINE configs[3] / [4]: the sharded top-k path is the timed step
        a.workload = "topk"
        a.users = -(-10_000_000 // world)
        a.dishes = 1_000_000
        a.embed = 64 if a.config == 3 else 128
    C, E, U, I, B = 4, a.embed, a.users, a.dishes, a.pairs
    user_base = rank * U
    PM, RE, CE, users, items, cats = make_inputs(torch, dev, U, I, C, E, B, 20260101 + 2 + rank, user_base)
    if a.unique_users:
        g = torch.Generator(device=dev); g.manual_seed(7)
        users = (torch.randperm(U, generator=g, device=dev)[:B].to(torch.int32) + int(user_base)).contiguous()
    eng = foodrec_amd.ScoringEngine(PM, RE, CE, coef=0.99, device=dev, user_base=user_base)
    for kv in a.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    out = torch.empty(B, dtype=torch.float32, device=dev)
    mlp = a.workload == "mlp"
    wl = a.workload
    g = torch.Generator(device=dev); g.manual_seed(20260101 + 3)          # same on every rank: replicated tables
    K = (C + 1) * E
    if wl in ("mlp", "topk"):
        pat = torch.randint(1, 2 ** C, (I,), generator=g, device=dev, dtype=torch.int32)
        dcat = ((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).float()
        if wl == "topk" and a.topk_weighted_masks:              # any float weight is legal input
            dcat = dcat * (0.5 + 1.5 * torch.rand((I, C), generator=g, device=dev))
        eng.set_dish_categories(dcat)
    mlp_head = mlp_cats = None
    if wl == "mlp":
        rn = lambda *shape: torch.randn(shape, generator=g, device=dev)
        mlp_head = (rn(K, 256) / K ** 0.5, rn(256) * 0.1, rn(256, 64) / 16.0, rn(64) * 0.1, rn(64) / 8.0, 0.0)
        eng.set_mlp_head(*mlp_head)
        mlp_cats = ((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).float()
    if wl == "ingredients" or (wl == "topk" and a.topk_with_ingredients):
        R = a.ingredients
        lens = torch.randint(1, 21, (I,), generator=g, device=dev)          # 1..20 ingredients per dish (build-chosen)
        off = torch.zeros(I + 1, dtype=torch.int32, device=dev)
        off[1:] = torch.cumsum(lens, 0).to(torch.int32)
        nnz = int(off[-1].item())
        eng.set_ingredients(torch.randn((R, E), generator=g, device=dev) * E ** -0.5, off,
                            torch.randint(0, R, (nnz,), generator=g, device=dev, dtype=torch.int32))
    tk_users = sharded = None
    round_cfg = default_round_users(U, a.round_users)                    # users per retrieval launch of the sharded top-k path
    if wl == "topk":
        from foodrec_amd.sharding import UserShardedScorer
        n_tk = U if a.config is not None else min(a.topk_users if a.topk_users > 0 else 65536, U)
        tk_users = (torch.randperm(U, generator=torch.Generator(device=dev).manual_seed(11 + rank), device=dev)[:n_tk]
                    .to(torch.int32) + int(user_base)).contiguous()
        sharded = UserShardedScorer(eng, world * U, device=dev)

    def step():
        if wl == "pairs":
            eng.score_pairs(users, items, cats, out=out)
        elif wl == "ingredients":
            eng.score_pairs_ingredients(users, items, cats, out=out)
        elif wl == "mlp":
            eng.score_pairs_mlp(users, items, out=out)
        elif a.config is not None:                                       # every user of the shard, rounds, ONE all-gather
            sharded.topk_all_users(10, round_users=round_cfg)
        else:                                                            # retrieval: per-shard top-k, then the exchange
            sharded.topk_users_gathered(tk_users, a.topk_k)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

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

    # SURVEY.md 8d's count charges every row of the user block (1 564 B per pair at E = 64).  The timed kernel leaves out the
    # rows of weight-0 categories, so that count cannot be applied to ITS time (it would price bytes that were not
    # moved): the same batch is timed once more with every row fetched (option skip_masked = 0, into a buffer of its own)
    # and the survey's formula is published from that time.
    survey_ms = None
    if rank == 0 and wl == "pairs" and opts_used["skip_masked"] != 0 and not a.no_side:
        so = torch.empty_like(out)
        eng.set_option("skip_masked", 0)
        try:
            time_steps(torch, eng, users, items, cats, so, 3)
            _, per_lit = time_steps(torch, eng, users, items, cats, so, max(10, min(a.steps, 50)))
            eng.check()
            survey_ms = sum(per_lit) / len(per_lit)
            survey_kernel = eng.last_kernel()
        finally:
            eng.set_option("skip_masked", opts_used["skip_masked"])
        del so

    rc = 0
    line = None
    if rank == 0:
        bpp_survey = algorithmic_bytes_per_pair(C, E)
        skip = wl in ("pairs", "ingredients") and eng.get_option("skip_masked") != 0
        mean_active = float((cats != 0).sum(1).float().mean().item()) if skip else float(C)
        bpp = algorithmic_bytes_per_pair(C, E, mean_active) if skip else bpp_survey
        avg_ms = sum(per_launch_ms) / len(per_launch_ms)
        achieved = bpp * B / (avg_ms * 1e-3) / 1e9
        traffic = traffic_probe = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = "E%d_B%d_U%d_I%d%s" % (E, B, U, I, "_skip" if skip else "")
                traffic = tj.get(key, {}).get("fabric_bytes_per_launch", tj.get(key, {}).get("hbm_bytes_per_launch"))
                traffic_probe = tj.get(key, {}).get("stream_probe_GBps")
            except Exception:
                traffic = None
        table_bytes = 4 * (PM.numel() + RE.numel())
        cache_resident = table_bytes <= INFINITY_CACHE_BYTES
        units = (tk_users.numel() * I) if wl == "topk" else B           # (user, dish) pairs scored per step per GPU
        line = {
            "metric": "scored (user,dish) pairs/sec", "value": world * units * a.steps / wall_max, "unit": "pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall_max / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic %d users x %d dishes per GPU, C=%d categories, "
                                   "E=%d, uniform random (user,dish) pairs with per-pair category masks; reference "
                                   "forward Model_Recommender.py:56-97 (ingredient table / MLP head are "
                                   "build-defined extensions, not in this step)%s" %
                                   (U, I, C, E, "; Personal_Memory rows of categories with mask weight 0 are not fetched -- "
                                    "the reference graph multiplies them by 0 (option skip_masked = 1, same scores to the bit)"
                                    if skip else ""),
                       "users_per_gpu": U, "dishes": I, "categories": C, "embed_size": E, "pairs_per_step_per_gpu": B,
                       "sharding": "user-range shard per GPU, dishes replicated, no data-path collective",
                       "kernel": kernel_used, "options": opts_used},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/traffic.json" if traffic is not None else None,
                         "traffic_kind": ("L2<->fabric bytes per launch (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE = TCC_EA0 request "
                                          "counters); they INCLUDE Infinity-Cache hits, so this bounds HBM bytes from above; a "
                                          "COUNTER READING OF ANOTHER RUN of this command (separate --pmc passes, "
                                          "scripts/profile_gpu.sh), read from the committed file and kept only while this box's "
                                          "streaming-read probe is within 5 % of the profiled box's") if traffic is not None else None,
                         "kernel_avg_ms": avg_ms, "algorithmic_bytes_per_pair": bpp, "pairs_per_launch": B,
                         "table_bytes": table_bytes},
        }
        line["config"]["roofline_frac_is"] = (
            "roofline.frac prices the bytes the timed kernel has to move: (2 + active categories) x E x 4 + C x 4 + 12 per "
            "pair, %.1f B on this batch -- the user block's rows of categories the dish does not have are multiplied by 0 in "
            "Model_Recommender.py:82 and are not fetched.  SURVEY.md 8d's formula charges all C rows (%d B per pair); it is "
            "published from a run that fetches them all: roofline.survey_8d_frac = %d B x pairs / roofline.survey_8d_ms / peak"
            % (bpp, bpp_survey, bpp_survey)) if skip else "roofline.frac follows SURVEY.md 8d: (C + 2) x E x 4 + C x 4 + 12 bytes per pair"
        if survey_ms is not None:
            line["roofline"].update({"survey_8d_ms": survey_ms, "survey_8d_bytes_per_pair": bpp_survey,
                                     "survey_8d_GBps": bpp_survey * B / (survey_ms * 1e-3) / 1e9,
                                     "survey_8d_frac": bpp_survey * B / (survey_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "survey_8d_pairs_per_s": B / (survey_ms * 1e-3),
                                     "survey_8d_kernel": survey_kernel + " (option skip_masked = 0: every row of the user block fetched)"})
        if skip:
            line["roofline"].update({
                "mean_active_categories": mean_active, "survey_bytes_per_pair": bpp_survey,
                "bytes_model": "mask-aware: (2 + active categories) x E x 4 + C x 4 + 12 per pair, averaged over the batch. "
                               "SURVEY.md 8d's count charges all C low-level rows of the user block (%d B); the rows of "
                               "categories whose mask weight is 0 are multiplied by 0 in the reference graph and this "
                               "kernel does not fetch them (option skip_masked, default 1), so by that count the same run "
                               "would read %.3f of the peak -- more bytes than were moved" %
                               (bpp_survey, bpp_survey * B / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)})

    # the headline is complete here; everything below decorates it.  Should a leg never return (a collective that
    # does not complete on some rank), rank 0 still prints the line and every rank leaves.
    in_flight = {"leg": "none"}

    def give_up():
        # a leg did not return (a collective that never completes on some rank, a hung launch): the headline is not lost,
        # but the run did NOT end cleanly -- exit status 4, and the record says which leg this rank was in
        msg = "rank %d: leg '%s' did not return within %.0f s" % (rank, in_flight["leg"], a.side_timeout)
        print("bench.py: " + msg, file=sys.stderr)
        sys.stderr.flush()
        if rank == 0:
            line["side_legs"] = {"status": "not finished: headline line only", "rank": rank, "leg_in_flight": in_flight["leg"],
                                 "timeout_s": a.side_timeout, "exit_status": 4}
            print(json.dumps(line))
            sys.stdout.flush()
        os._exit(4)
    watchdog = threading.Timer(a.side_timeout, give_up)
    watchdog.daemon = True
    watchdog.start()

    # N > 1: the user-sharded retrieval step and the owner-routed pair step (outside the timed region, repeated)
    topk_ag = routed = scaling = None
    if not a.no_side and wl == "pairs":
        if a.topk_users > 0:
            in_flight["leg"] = "sharded_topk_allgather"
            try:
                topk_ag = sharded_topk_leg(torch, dist if use_dist else None, eng, U, I, C, E, dev, user_base,
                                           min(a.topk_users, U), world)
            except Exception as e:                                     # noqa: BLE001 -- never lose the headline line
                topk_ag = {"error": "%s: %s" % (type(e).__name__, e)}
        if use_dist:
            in_flight["leg"] = "routed_pairs_alltoall"
            try:
                routed = routed_pairs_leg(torch, dist, eng, U, I, C, dev, world, min(B, 1 << 22))
            except Exception as e:                                     # noqa: BLE001
                routed = {"error": "%s: %s" % (type(e).__name__, e)}
        if a.scaling_users > 0:
            in_flight["leg"] = "scaling_path"
            try:
                scaling = scaling_path_block(torch, dist if use_dist else None, foodrec_amd, dev, world, rank, a.scaling_users,
                                             1_000_000, 64, 10, default_round_users(-(-a.scaling_users // world), a.round_users))
            except Exception as e:                                     # noqa: BLE001
                scaling = {"error": "%s: %s" % (type(e).__name__, e)}
            if world == 1 and rank == 0 and not a.no_projection and "error" not in scaling and a.scaling_users >= 8:
                in_flight["leg"] = "scaling_path.projected_world8"
                try:
                    scaling["projected_world8"] = projected_world8_block(torch, foodrec_amd, dev, a.scaling_users, 1_000_000, 64, 10,
                                                                         scaling.get("wall_ms"))
                except Exception as e:                                 # noqa: BLE001
                    scaling["projected_world8"] = {"error": "%s: %s" % (type(e).__name__, e)}
    in_flight["leg"] = "rank-0 side measurements"

    if a.sweep and rank == 0:
        so = torch.empty_like(out)
        for pf in (1, 2, 4):
            for nt in (0, 1):
                for bpc in (2, 4, 8, 16):
                    eng.set_option("prefetch", pf); eng.set_option("nt_loads", nt); eng.set_option("blocks_per_cu", bpc)
                    time_steps(torch, eng, users, items, cats, so, 3)
                    w, per = time_steps(torch, eng, users, items, cats, so, 10)
                    ms = sorted(per)[len(per) // 2]
                    print("sweep pf=%d nt=%d blocks_per_cu=%2d: %.3f ms  %.2f Gpairs/s  %.0f GB/s" %
                          (pf, nt, bpc, ms, B / ms / 1e6, B * (line["roofline"]["algorithmic_bytes_per_pair"] if line else
                                                            algorithmic_bytes_per_pair(C, E)) / ms / 1e6), file=sys.stderr)
        for k, v in opts_used.items():
            eng.set_option(k, v)

    if rank == 0:
        if cache_resident and wl in ("pairs", "ingredients"):
            # both tables stay in the 256 MiB Infinity Cache between launches: the algorithmic rate is a cache rate and can
            # exceed the HBM peak, so no HBM fraction is published for this shape
            line["roofline"].update({"bound": "cache", "peak": None, "frac": None,
                                     "note": "tables (%.0f MB) fit the 256 MiB Infinity Cache: rows are re-read on-die, the HBM "
                                             "roofline does not bound this shape" % (table_bytes / 1e6)})
        if mlp:
            K = (C + 1) * E
            fl = 2.0 * (K * 256 + 256 * 64 + 64)
            tf = fl * B / (avg_ms * 1e-3) / 1e12
            line["config"]["workload"] = ("BASELINE configs[2]: synthetic %d users x %d dishes per GPU, C=%d, E=%d + "
                                          "BUILD-DEFINED 3-layer head %d->256->64->1 on the interaction vector (no "
                                          "reference counterpart; parity vs the build's own restatement only); "
                                          "uniform random pairs, masks from the resident dish table" % (U, I, C, E, K))
            x3 = kernel_used.endswith("bf16x3")
            # split-bf16 form: layers 1-2 run as 3 bf16 MFMAs per product (executed flops = 3 x algorithmic) against the
            # dense bf16 peak; the exact form runs everything on the f32 MFMA against its peak.  The producer / consumer
            # kernel groups the pairs by dish mask pattern and runs only the k-blocks a pattern keeps (the E k-values of a
            # category of weight 0 are zeros in z): executed flops and fetched bytes count those blocks only
            grouped = kernel_used.startswith("m2d_mlp_pc") and eng.get_option("skip_masked") != 0 and E >= 64
            act = float((mlp_cats[items.long()] != 0).sum(1).float().mean().item()) if grouped else float(C)
            Ka = (1.0 + act) * E                                          # k-values of layer 1 actually multiplied, per pair
            ex = 3.0 * 2.0 * (Ka * 256 + 256 * 64) * B / (avg_ms * 1e-3) / 1e12 if x3 else tf
            dense_ex = 3.0 * 2.0 * (K * 256 + 256 * 64) * B / (avg_ms * 1e-3) / 1e12 if x3 else tf
            peak = 2500.0 if x3 else 157.3
            hbm = (2 * Ka * 4 + 12) * B / (avg_ms * 1e-3) / 1e9
            line["roofline"] = {"bound": "mfma", "achieved": ex, "peak": peak, "unit": "TFLOP/s", "frac": ex / peak,
                                "mean_active_categories": act, "k_values_multiplied_per_pair": Ka,
                                "dense_equivalent_frac": dense_ex / peak,
                                "traffic": None, "kernel_avg_ms": avg_ms, "flop_per_pair": fl, "pairs_per_launch": B,
                                "algorithmic_tflops": tf, "f32_mfma_equivalent_frac": tf / 157.3,
                                "dtype": ("split bf16 for layers 1-2 (3 x %s per product, fp32 accumulate)"
                                          % ("v_mfma_f32_16x16x32_bf16" if kernel_used.startswith("m2d_mlp_pc") else "v_mfma_f32_32x32x16_bf16")
                                          if x3 else "f32 (v_mfma_f32_32x32x2_f32, exact)"),
                                "hbm_algorithmic_GBps": hbm, "hbm_frac": hbm / HBM_PEAK_GBS}
            if x3:
                line["roofline"].update(bare_loop_fields(ex))
            line["dtype"] = "bf16x3" if x3 else "f32"
            if not a.no_cpu_baseline and world == 1:
                cb, ok = mlp_baseline(torch, PM, RE, CE, mlp_cats, mlp_head, users, items, user_base, mlp_sample, a.cpu_seconds)
                line["cpu_baseline"] = cb
                if not ok:
                    rc = 3
        if wl == "ingredients":
            bpp_i = (3.0 + mean_active) * E * 4 + C * 4 + 12            # one extra E-float row per pair
            ach = bpp_i * B / (avg_ms * 1e-3) / 1e9
            line["config"]["workload"] = ("BASELINE configs[1] WITH the build-defined ingredient table: %d users x %d dishes "
                                          "x %d ingredients per GPU, 1-20 ingredients per dish, E=%d; high-level path from the "
                                          "per-dish multi-hot ingredient sum (hoisted to a per-table segment-sum kernel), "
                                          "low-level path and blend as Model_Recommender.py:82-96; no reference counterpart"
                                          % (U, I, a.ingredients, E))
            line["roofline"].update({"achieved": ach, "algorithmic_bytes_per_pair": bpp_i, "traffic": None, "traffic_kind": None})
            if not cache_resident:
                line["roofline"]["frac"] = ach / HBM_PEAK_GBS
        if wl == "topk":
            x3 = kernel_used.endswith("bf16x3")
            Ew = 2 * E if a.topk_with_ingredients else E                 # grouped rows are [H[d] | RE[d]] with the ingredient table
            fl = (2.0 * Ew * (3 if x3 else 1) if kernel_used.startswith("m2d_topk_grouped") else 2.0 * K) * units
            # the pattern-grouped kernels step through their blocks' relevant mask patterns only: executed flops = that share
            # of the catalogue's (the share of the step's last launch stands for the step)
            scanned_frac = None
            if kernel_used.startswith("m2d_topk_grouped") and not a.topk_with_ingredients:
                sc_, fu_ = eng.get_option("topk_tiles_scanned"), eng.get_option("topk_tiles_full")
                if sc_ > 0 and fu_ > 0:
                    scanned_frac = sc_ / fu_
            tf_all = fl / (avg_ms * 1e-3) / 1e12
            tf = tf_all * (scanned_frac if scanned_frac is not None else 1.0)
            # the hi x hi first form (large catalogues): one product per tile stepped through, the two cross
            # products for the share of (wave, tile) pairs that could still hold a candidate (the step's last launch stands for the step)
            cross_share = None
            if x3 and kernel_used.startswith("m2d_topk_grouped"):
                cmp_ = eng.get_option("topk_tiles_completed")
                sc2_ = eng.get_option("topk_tiles_scanned")              # (with the ingredient table: every tile, no pattern is pruned)
                if cmp_ >= 0 and sc2_ > 0:
                    cross_share = cmp_ * 32.0 / (sc2_ * eng.get_option("topk_block_users"))
                    tf = tf / 3.0 * (1.0 + 2.0 * cross_share)
            peak = 2500.0 if x3 else 157.3
            line["config"]["workload"] = (("BASELINE configs[%d]: %d users over %d GPU(s) (%d per GPU) x %d replicated dishes, E=%d: "
                                           "full-catalogue top-10 for EVERY user of the shard in rounds of %d, then ONE all-gather of "
                                           "[shard,10] x (f32 score, i32 id) (%d bytes per rank); build-defined generalisation of "
                                           "evaluate.py:39-63" % (a.config, world * U, world, U, I, E, round_cfg, U * 80))
                                          if a.config is not None else
                                          "BASELINE configs[3]/[4] retrieval: full-catalogue top-%d for %d users per GPU over %d "
                                          "replicated dishes (users from this GPU's %d-user shard), E=%d, then all-gather of "
                                          "[users,%d] x (f32 score, i32 id); build-defined generalisation of evaluate.py:39-63"
                                          % (a.topk_k, tk_users.numel(), I, U, E, a.topk_k)) + (
                " -- WEIGHTED category masks (any float is legal placeholder input, Model_Recommender.py:32): no pattern grouping, "
                "the dense exact-f32 kernel contracts over (C + 1) E" if a.topk_weighted_masks else "") + (
                " -- WITH the build-defined ingredient table (%d rows, 1-20 per dish)" % a.ingredients if a.topk_with_ingredients else "")
            line["roofline"] = {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                                "traffic": None, "step_avg_ms": avg_ms, "flop_per_pair_executed": (fl / units if cross_share is None else fl / units / 3.0 * (1.0 + 2.0 * cross_share)),
                                "dense_equivalent_tflops": 2.0 * K * units / (avg_ms * 1e-3) / 1e12,
                                "scanned_fraction": scanned_frac, "frac_if_every_tile_were_scanned": tf_all / peak,
                                "hi_first_form": cross_share is not None, "share_of_wave_tiles_given_cross_products": cross_share,
                                # the same tiles priced as the three-product form would execute them: the rate the hi x hi first form is worth
                                "frac_if_priced_as_three_products": (tf * 3.0 / (1.0 + 2.0 * cross_share) / peak if cross_share is not None else None),
                                "note": "`frac` prices the flops EXECUTED (the tiles the blocks stepped through), and `value` counts the pairs "
                                        "of those tiles; pairs_decided_per_s counts every (user, dish) pair of the catalogue -- most are "
                                        "decided by a bound, without being multiplied",
                                "dtype": (("split bf16: hi x hi for every tile, lo x hi + hi x lo for the tiles that can hold a candidate "
                                           "(v_mfma_f32_32x32x16_bf16, fp32 accumulate)" if cross_share is not None else
                                           "split bf16 (3 x v_mfma_f32_32x32x16_bf16, fp32 accumulate)") if x3 else
                                          "f32 (v_mfma_f32_32x32x2_f32, exact)")}
            if x3:
                line["roofline"].update(bare_loop_fields(tf))
            line["dtype"] = "bf16x3" if x3 else "f32"
            # "scored" is claimed only for the pairs that were multiplied: the tiles the blocks stepped through.  Every pair
            # of the catalogue is DECIDED (ranked or excluded by a bound) at the rate beside it.
            decided = line["value"]
            line["pairs_decided_per_s"] = decided
            line["value"] = decided * (scanned_frac if scanned_frac is not None else 1.0)
            line["pairs_multiplied_per_s"] = line["value"]
            line["value_definition"] = ("v2 (rounds 4+): pairs MULTIPLIED per second; the BENCH / profiles records of rounds 1-3 "
                                        "published what is now pairs_decided_per_s under `value`")
            line["value_is"] = ("(user, dish) pairs multiplied per second, whole job: every pair of the catalogue is decided at "
                                "pairs_decided_per_s, the share `roofline.scanned_fraction` of them by being scored -- the others by "
                                "a bound on their mask pattern's scores (DESIGN.md 4.4)")
            if sharded is not None and getattr(sharded, "last_allgather_events", None):
                line["allgather_exposed_ms"] = sharded.last_allgather_events[0].elapsed_time(sharded.last_allgather_events[1])
            line["roofline"]["allgather_bytes_per_rank"] = tk_users.numel() * 8 * a.topk_k if use_dist else 0
            line["roofline"]["repaired_users_last_launch"] = eng.get_option("topk_repaired")
        if not a.no_side and wl == "pairs":
            in_flight["leg"] = "no-reuse / stream probe"
            nr, probe, hbm_only = side_measurements(torch, eng, PM, U, I, C, E, dev, user_base)
            line["roofline"]["no_reuse"] = nr
            line["roofline"]["hbm_only"] = hbm_only
            line["roofline"]["stream_read_probe"] = probe
            line["roofline"]["frac_of_stream_probe"] = achieved / probe["GBps"]
            # the same figures as scalars of `roofline` (a record that keeps only scalar keys keeps these)
            line["roofline"].update({
                "stream_probe_GBps": probe["GBps"],
                "no_reuse_GBps": nr["achieved"], "no_reuse_frac": nr["frac"],
                "hbm_only_GBps": hbm_only["achieved"], "hbm_only_frac_of_spec": hbm_only["frac_of_spec_peak"],
                "hbm_only_frac_of_stream_probe": hbm_only["frac_of_stream_probe"],
                "hbm_only_masked_GBps": hbm_only["masked"]["achieved"],
                "hbm_only_masked_frac_of_spec": hbm_only["masked"]["frac_of_spec_peak"],
                "hbm_only_masked_frac_of_stream_probe": hbm_only["masked"]["frac_of_stream_probe"]})
            if traffic is not None:
                # the counter reading belongs to the box it was taken on: kept only while this box streams like that one
                line["roofline"]["traffic_profiled_box_stream_probe_GBps"] = traffic_probe
                if traffic_probe is None or abs(probe["GBps"] / traffic_probe - 1.0) > 0.05:
                    line["roofline"].update({"traffic": None, "traffic_dropped": "this box's stream probe (%.0f GB/s) is not within 5 %% "
                                             "of the profiled box's (%s GB/s): the committed counter reading is not published for it"
                                             % (probe["GBps"], "%.0f" % traffic_probe if traffic_probe else "unrecorded")})
        if a.topk_users > 0 and not a.no_side and wl == "pairs":
            in_flight["leg"] = "catalogue_topk"
            lists_x3, lists_f32 = {}, {}
            line["catalogue_topk"] = catalogue_topk_leg(torch, eng, U, I, C, E, dev, user_base, min(a.topk_users, U), keep=lists_x3)
            eng.set_option("topk_bf16x3", 0)            # the exact-f32 kernel's figure beside the split-bf16 one (same users, same dishes)
            try:
                line["catalogue_topk"]["exact_f32"] = catalogue_topk_leg(torch, eng, U, I, C, E, dev, user_base, min(a.topk_users, U),
                                                                         keep=lists_f32)
            finally:
                eng.set_option("topk_bf16x3", 1)
            line["catalogue_topk"]["index_exactness"] = compare_lists(torch, lists_x3, lists_f32)
            line["catalogue_topk"]["index_exactness"].update({"refined_users": eng.get_option("topk_refined"),
                                                               "refined_users_sent_to_the_repair": eng.get_option("topk_refine_repaired")})
            del lists_x3, lists_f32
            eng.set_option("topk_refine", 0)            # ... and what finishing the near-tied lists in one arithmetic costs (same call without it)
            try:
                lists_off = {}
                off = catalogue_topk_leg(torch, eng, U, I, C, E, dev, user_base, min(a.topk_users, U), keep=lists_off)
                line["catalogue_topk"]["refine_off"] = {"median_ms": off["median_ms"], "pairs_per_s": off["pairs_per_s"],
                                                        "what": "option topk_refine = 0: the split-bf16 lists as the scan leaves them (round 3's "
                                                                "behaviour); ids then differ from the exact-f32 kernel's wherever two scores sit inside "
                                                                "the split's rounding"}
                eng.set_option("topk_prune", 0)         # the every-tile form without it: the scan's own matrix-pipe fraction
                try:
                    et = catalogue_topk_leg(torch, eng, U, I, C, E, dev, user_base, min(a.topk_users, U))
                    line["catalogue_topk"]["refine_off"]["every_tile"] = {"median_ms": et["median_ms"], "roofline_frac": et["roofline"]["frac"]}
                finally:
                    eng.set_option("topk_prune", 1)
                eng.set_option("topk_bf16x3", 0)
                lists_f32_off = {}
                catalogue_topk_leg(torch, eng, U, I, C, E, dev, user_base, min(a.topk_users, U), keep=lists_f32_off)
                line["catalogue_topk"]["refine_off"]["index_exactness"] = compare_lists(torch, lists_off, lists_f32_off)
                del lists_off, lists_f32_off
            finally:
                eng.set_option("topk_bf16x3", 1)
                eng.set_option("topk_refine", 1)
            eng.set_option("topk_prune", 0)             # ... and the same kernel made to step through every tile: the MFMA-bound form
            try:
                line["catalogue_topk"]["every_tile"] = catalogue_topk_leg(torch, eng, U, I, C, E, dev, user_base, min(a.topk_users, U))
            finally:
                eng.set_option("topk_prune", 1)
        if world == 1 and not a.no_side and wl == "pairs" and not a.no_cpu_baseline:
            in_flight["leg"] = "evaluator"
            try:
                line["evaluator"] = evaluator_leg(torch, dev)
            except Exception as e:                                     # noqa: BLE001
                line["evaluator"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not a.no_side and wl == "pairs":
            in_flight["leg"] = "with_user_high_table"
            try:
                line["with_user_high_table"] = user_high_leg(torch, eng, users, items, cats, C, E)
            except Exception as e:                                     # noqa: BLE001
                line["with_user_high_table"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not a.no_side and wl == "pairs":
            in_flight["leg"] = "with_ingredient_table"
            try:
                line["with_ingredient_table"] = ingredients_leg(torch, eng, users, items, cats, I, C, E, dev, a.ingredients)
            except Exception as e:                                     # noqa: BLE001
                line["with_ingredient_table"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if scaling is not None:
            if world > 1 and wl == "pairs" and "error" not in scaling:
                # a SCALE record headlines `value`, which is the collective-free pair path in weak scaling (about N x by
                # construction): say first where the path north_star scales is
                line["config"]["workload"] = ("[topk_path_ms = %.1f: the STRONGLY scaled user-sharded top-k path (%d users over %d "
                                              "GPUs x %d dishes, per-round all-gathers) is in topk_path_* / scaling_path -- `value` below "
                                              "is the weakly scaled pair path] " % (scaling.get("wall_ms") or float("nan"),
                                                                                   scaling.get("users_total") or 0, world,
                                                                                   scaling.get("dishes") or 0)) + line["config"]["workload"]
            line["scaling_path"] = scaling                              # ("scaling" itself is the contract's "weak" / "strong" string)
            # the same as scalars of the line itself (a record that keeps only top-level scalar keys keeps these): the path
            # north_star's ">= 6x at 8 GPUs" speaks of
            line.update({"topk_path_ms": scaling.get("wall_ms"),
                         "topk_path_pairs_decided_per_s": scaling.get("pairs_decided_per_s_whole_job"),
                         "topk_path_pairs_multiplied_per_s": scaling.get("pairs_multiplied_per_s_whole_job"),
                         "topk_path_allgather_exposed_ms": scaling.get("allgather_exposed_ms"),
                         "topk_path_dtype": "bf16x3" if str(scaling.get("kernel", "")).endswith("bf16x3") else "f32",
                         "topk_path_users_total": scaling.get("users_total"), "topk_path_dishes": scaling.get("dishes")})
            pj = scaling.get("projected_world8") or {}
            if "shard_ms" in pj:                                        # (one-GPU projection of N = 8: labelled as such)
                line.update({"topk_path_projected_world8_shard_ms": pj["shard_ms"],
                             "topk_path_projected_world8_speedup_upper_bound": pj["implied_speedup_upper_bound"],
                             "topk_path_projected_world8_status": pj["status"]})
        if topk_ag is not None:
            line["sharded_topk_allgather"] = topk_ag
        if routed is not None:
            line["routed_pairs_alltoall"] = routed
        if a.unique_users:
            line["config"]["workload"] += " [--unique-users: every user at most once per step]"
        in_flight["leg"] = "cpu_baseline"
        if world == 1 and not a.no_cpu_baseline and wl == "pairs":
            cb, ref, _ = cpu_baseline(torch, PM, RE, CE, users, items, cats, a.cpu_seconds)
            # the baseline doubles as a live parity check of the TIMED kernel's output (sampled right after the timed
            # region, before any side leg ran) on the same pairs
            got = timed_sample.cpu()
            err = ((got - ref).abs() / ref.abs().clamp(min=1.0)).max().item()
            cb["max_abs_diff_vs_gpu"] = (got - ref).abs().max().item()
            cb["max_rel_diff_vs_gpu"] = err
            cb["parity_tolerance"] = PARITY_TOL
            cb["parity_ok"] = bool(err <= PARITY_TOL and torch.equal(torch.isnan(got), torch.isnan(ref)))
            line["cpu_baseline"] = cb
            if not cb["parity_ok"]:
                rc = 3
                print("bench.py: PARITY FAILURE: timed kernel vs CPU restatement, max |d| / max(1, |ref|) = %.3e > %.0e"
                      % (err, PARITY_TOL), file=sys.stderr)
        if rehearse:
            line["rehearsal"] = ("%d ranks share ONE GPU, collectives over gloo: a functional run of the N > 1 path "
                                 "(M2D_BENCH_REHEARSE_ONE_GPU=1); its timings and rates mean nothing" % world)
        watchdog.cancel()
        print(json.dumps(line))
        sys.stdout.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    watchdog.cancel()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
