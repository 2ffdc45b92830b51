"""User-sharded legs: per-shard top-k + all-gather, pairs routed to their owners, the `scaling_path`
block at BASELINE configs[3] / [4]'s per-GPU shape, and the identity of the world an N > 1 line was measured on."""
from __future__ import annotations

import os
import socket
import time

from .common import (BF16_MFMA_PEAK_TFLOPS, F32_MFMA_PEAK_TFLOPS, XGMI_LINK_GBS, Clock, masks_from_patterns,
                     median, random_masks)


def default_round_users(per_gpu_users, requested):
    """--round-users 0 (the default): the shard cut into the fewest EVEN rounds of at most 524 288 users -- one round
    of 500 000 at N = 1's 10 M users (20 of them), three of 416 667 at N = 8's 1.25 M (a short last round pays a
    retrieval call's fixed launches for a fraction of the work)."""
    if requested:
        return int(requested)
    if per_gpu_users <= 0:
        return 524288
    return -(-per_gpu_users // -(-per_gpu_users // 524288))


def _device_identity(torch, dev):
    """A string that differs between two physical GPUs of one node and is the same for two ranks on one GPU."""
    dev = torch.device(dev)
    if dev.type != "cuda":
        return "cpu:%s:%d" % (socket.gethostname(), os.getpid())
    props = torch.cuda.get_device_properties(dev)
    for attr in ("uuid", "pci_bus_id"):
        v = getattr(props, attr, None)
        if v not in (None, ""):
            extra = getattr(props, "pci_device_id", "")
            return "%s:%s:%s" % (attr, v, extra)
    return "index:%d" % (dev.index or 0)


def world_identity(torch, dist, dev, world, rank):
    """What lets a reader of an N > 1 line verify that N ranks on N distinct devices took part: `ranks_seen` is an
    all-reduce of 1 over the job's process group, `distinct_devices` counts the distinct device identities
    (uuid / PCI bus id) all-gathered from the ranks.  `dist` is None in a single-process run."""
    me = _device_identity(torch, dev)
    backend, ids, seen = "none (single process)", [me], 1
    if dist is not None:
        backend = str(dist.get_backend())
        one = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        seen = int(one.item())
        ids = [None] * world
        dist.all_gather_object(ids, me)
    rccl = None
    try:
        if torch.device(dev).type == "cuda":
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:                                                   # noqa: BLE001 -- a label, never fatal
        rccl = None
    return {"world_size": world, "ranks_seen": seen, "distinct_devices": len(set(ids)), "device_ids": ids,
            "backend": backend, "rccl_version": rccl, "hip": getattr(torch.version, "hip", None),
            "host": socket.gethostname(),
            "what": "ranks_seen = all-reduce(SUM) of 1; device_ids = all-gather of each rank's device uuid / PCI bus "
                    "id; a healthy N-GPU run has ranks_seen == distinct_devices == n_gpus"}


def _per_rank(torch, dist, dev, world, values):
    """All-gather a few float64 per rank -> list (rank-major) of lists."""
    t = torch.tensor(values, dtype=torch.float64, device=dev)
    if dist is None:
        return [t.tolist()]
    out = torch.empty(world * len(values), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, t)
    return out.view(world, len(values)).tolist()


def sharded_topk_leg(torch, dist, eng, U, I, C, E, dev, user_base, n_users, world, k=10, repeats=7):
    """Every rank: top-k over the replicated catalogue for n_users of ITS users, then ONE all-gather of
    [n_users, k] x (f32 score, i32 id) per rank, through foodrec_amd.sharding.  Timed
    `repeats` times between barriers; the median of the max-over-ranks wall time is reported.  `dist` is None in a
    single-process run: the same leg with no peers, so that the N = 1 line carries the number the N > 1 lines are
    compared with."""
    from foodrec_amd.sharding import UserShardedScorer
    g = torch.Generator(device=dev)
    g.manual_seed(11)                                     # same dish masks on every rank (replicated)
    _, dish_cats = random_masks(torch, I, C, dev, g)
    eng.set_dish_categories(dish_cats)
    sh = UserShardedScorer(eng, world * U, device=dev, always_collective=dist is not None)
    users = (torch.randperm(U, generator=g, device=dev)[:n_users].to(torch.int32) + int(user_base)).contiguous()
    sh.topk_users_gathered(users[:1024], k)               # builds the retrieval tables, warms RCCL up
    for _ in range(5):                                    # the first full launches run 5-10 % slow (clock ramp)
        sh.topk_users_gathered(users, k)
    walls, tk_ms, ag_ms = [], [], []
    clk = Clock(torch, dev)
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
        walls.append(w)
        tk_ms.append(a_)
        ag_ms.append(b_)
    eng.check()
    r = dist.get_rank() if dist is not None else 0
    mine = slice(r * n_users, (r + 1) * n_users)
    ok = bool(torch.equal(gi[mine], ids) and torch.equal(gs[mine], s))
    wall = median(walls)
    return {"users_per_gpu": n_users, "dishes": I, "k": k, "repeats": repeats, "wall_ms_median": wall * 1e3,
            "wall_ms_all": [w * 1e3 for w in walls], "topk_ms_median": median(tk_ms),
            "allgather_ms_median": median(ag_ms),
            "allgather_bytes_per_rank": n_users * k * 8 if dist is not None else 0,
            "users_per_s_whole_job": world * n_users / wall, "pairs_per_s_whole_job": world * n_users * I / wall,
            "kernel": eng.last_kernel(), "own_slice_roundtrip_ok": ok,
            "what": "max over ranks per repeat, median over repeats; per-shard full-catalogue top-k + one RCCL "
                    "all-gather" + ("" if dist is not None else " (single process: no peers, no collective)")}


def world_rows_ok(torch, sh, gs, gi):
    """The gathered result's padding: rows of rank r beyond its shard's count do not exist (the result is trimmed to
    the users that do), and every existing row holds dish ids >= 0 -- a cheap check of the OTHER ranks' slices (their
    content is checked by the rank that owns them)."""
    if gi.shape[0] != sh.num_users_total:
        return False
    return bool((gi >= 0).all())


def sharded_all_users_leg(torch, dist, sh, I, k, round_users, repeats=1, warm_rounds=2):
    """The user-sharded top-k path as north_star states it: every rank ranks EVERY user of its shard over the
    replicated catalogue in rounds of `round_users` users and the ranks exchange their final lists -- [shard, k] x
    (f32 score, i32 id) per rank -- by all-gather, one piece per round, each issued asynchronously while the next
    round is being ranked (UserShardedScorer.topk_all_users): only the last round's exchange is exposed.  `dist` is
    None in a single-process run (no peers, no collective).  Wall time = max over ranks, median over repeats;
    `allgather_exposed_ms` = what the stream still waited for after the last round's kernels.  Per rank (so that an
    N > 1 record shows a straggler): `shard_ms_per_rank`, `allgather_exposed_ms_per_rank` of the median repeat."""
    clk = Clock(torch, sh.device)
    per_round = min(int(round_users), max(sh.count, 1))
    first = torch.arange(sh.base, sh.base + min(per_round, sh.count), dtype=torch.int32, device=sh.device)
    if sh.count:
        sh.topk_local(k, first)                            # builds the retrieval tables
        for _ in range(warm_rounds):
            sh.topk_local(k, first)
    if dist is not None:                                  # the collective's buffers and connections, at real sizes
        sh.topk_all_users(k, round_users=round_users)
    walls, exposed, per_rank = [], [], []
    ok = True
    world = sh.world
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
        mine_wall = time.perf_counter() - t0
        ex = 0.0
        if dist is not None and getattr(sh, "last_allgather_events", None):
            ex = sh.last_allgather_events[0].elapsed_time(sh.last_allgather_events[1])
        rows = _per_rank(torch, dist, sh.device, world, [mine_wall, ex])
        walls.append(max(r[0] for r in rows))
        exposed.append(max(r[1] for r in rows))
        per_rank.append(rows)
    if sh.count:
        # Outside the timed region: this rank's WHOLE shard ranked again on its own, without any collective, must sit
        # in the gathered result bit for bit -- every round, the buffer-reusing ones (index >= 2) and a short last one
        ls, li = sh.topk_local_rounds(k, round_users)
        lo = sh.rank * sh.per if dist is not None else 0
        ok = bool(torch.equal(gi[lo:lo + sh.count], li) and
                  torch.equal(gs[lo:lo + sh.count].view(torch.int32), ls.view(torch.int32)))
        if dist is not None and world_rows_ok(torch, sh, gs, gi) is False:
            ok = False
    if sh.scorer is not None:
        sh.scorer.check()
    wall = median(walls)
    mid = per_rank[walls.index(wall)]
    total_users = sh.num_users_total
    ex_ranks = [r[1] for r in mid]
    return {"path": "sharded_topk_allgather", "users_total": total_users, "users_per_gpu": sh.per, "dishes": I,
            "k": k, "round_users": int(round_users), "rounds_per_gpu": -(-sh.per // int(round_users)),
            "repeats": repeats, "wall_ms": wall * 1e3,
            "allgather_exposed_ms": median(exposed) if dist is not None else 0.0,
            "shard_ms_per_rank": [r[0] * 1e3 for r in mid],
            "allgather_exposed_ms_per_rank": ex_ranks if dist is not None else [0.0],
            "allgather_exposed_ms_max": max(ex_ranks) if dist is not None else 0.0,
            "allgather_exposed_ms_min": min(ex_ranks) if dist is not None else 0.0,
            "allgather": ("one asynchronous all-gather per round, overlapped with the next round's ranking; exposed = "
                          "the last round's exchange and its copy into the result") if dist is not None
            else "none (single process)",
            "allgather_bytes_per_rank": sh.per * k * 8 if dist is not None else 0,
            "users_per_s_whole_job": total_users / wall, "pairs_per_s_whole_job": total_users * I / wall,
            "own_slice_roundtrip_ok": ok}


def _replicated_tables(torch, dev, I, C, E):
    """The replicated tables of the scaling_path blocks: the same on every rank."""
    g = torch.Generator(device=dev)
    g.manual_seed(20260101 + 4)
    sc = E ** -0.5
    RE = torch.randn((I, E), generator=g, device=dev) * sc
    CE = torch.randn((C, E), generator=g, device=dev) * sc
    pat = torch.randint(1, 2 ** C, (I,), generator=g, device=dev, dtype=torch.int32)
    return g, RE, CE, masks_from_patterns(torch, pat, C, dev)


def scaling_path_block(torch, dist, foodrec_amd, dev, world, rank, users_total, I, E, k, round_users, repeats=1):
    """`scaling_path`: the sharded top-k path at BASELINE configs[3] / configs[4]'s per-GPU shape, on tables of its
    own (users_total / world users per GPU x I replicated dishes).  The split-bf16 kernel (the default) over every
    user of the shard + the all-gather; the exact-f32 kernel's rate beside it, on one round of users per GPU."""
    from foodrec_amd.sharding import UserShardedScorer, shard_range
    C = 4
    base, count = shard_range(users_total, world, rank)
    g, RE, CE, dish_cats = _replicated_tables(torch, dev, I, C, E)
    g.manual_seed(20260101 + 40 + rank)
    PM = torch.randn((max(count, 1), C + 1, E), generator=g, device=dev) * E ** -0.5
    eng = foodrec_amd.ScoringEngine(PM, RE, CE, coef=0.99, device=dev, user_base=base)
    eng.set_dish_categories(dish_cats)
    sh = UserShardedScorer(eng, users_total, device=dev, always_collective=dist is not None)
    out = sharded_all_users_leg(torch, dist, sh, I, k, round_users, repeats=repeats)
    out["kernel"] = eng.last_kernel()
    x3 = out["kernel"].endswith("bf16x3")
    out["dtype"] = "bf16x3 (x = hi + lo, 3 x v_mfma_f32_32x32x16_bf16, fp32 accumulate)" if x3 else "f32"
    out["embed_size"] = E
    flop = 2.0 * E * (3 if x3 else 1)                      # per (user, dish) on the pattern-grouped kernels
    peak = BF16_MFMA_PEAK_TFLOPS if x3 else F32_MFMA_PEAK_TFLOPS
    out["roofline_frac_of_mfma_peak"] = flop * out["pairs_per_s_whole_job"] / world / 1e12 / peak
    out["repaired_users_last_round"] = eng.get_option("topk_repaired")
    if x3:
        sc_, fl_ = eng.get_option("topk_tiles_scanned"), eng.get_option("topk_tiles_full")
        out["scanned_fraction_last_round"] = sc_ / fl_ if fl_ else None
        if fl_:
            out["roofline_frac_of_mfma_peak"] *= sc_ / fl_
        out["roofline_note"] = ("fraction of the dense bf16 MFMA peak on the flops executed (tiles stepped through x 3 "
                                "MFMAs); pairs_per_s_whole_job counts every (user, dish) pair of the catalogue")
    # pairs DECIDED (every pair of the catalogue: most by a bound, without being multiplied) and pairs MULTIPLIED (the
    # tiles the blocks stepped through; the last round's share stands for the shard)
    out["pairs_decided_per_s_whole_job"] = out["pairs_per_s_whole_job"]
    out["pairs_multiplied_per_s_whole_job"] = (out["pairs_per_s_whole_job"]
                                               * (out.get("scanned_fraction_last_round") or 1.0))
    # the exact-f32 kernel on one round of this shard's users (every rank at once; max over ranks)
    clk = Clock(torch, dev)
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
            a = clk.mark()
            eng.topk_users(users, k)
            b = clk.mark()
            clk.sync()
            t = torch.tensor([clk.ms(a, b)], dtype=torch.float64, device=dev)
            if dist is not None:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms.append(float(t.item()))
        eng.check()
    if ms:
        m = median(ms)
        sc_, fl_ = eng.get_option("topk_tiles_scanned"), eng.get_option("topk_tiles_full")
        part = sc_ / fl_ if (fl_ and sc_) else 1.0           # tiles stepped through / all tiles (pattern pruning)
        out["exact_f32"] = {
            "kernel": eng.last_kernel(), "users_per_gpu_in_sample": n1, "topk_ms": m,
            "pairs_per_s_whole_job": world * n1 * I / m * 1e3,
            "roofline_frac_of_f32_mfma_peak": part * 2.0 * E * n1 * I / m / 1e9 / F32_MFMA_PEAK_TFLOPS,
            "scanned_fraction": part,
            "what": "option topk_bf16x3 = 0 (v_mfma_f32_32x32x2_f32, exact): one round of users per GPU, all ranks at "
                    "once, no all-gather; whole-shard time = this rate x the shard"}
    out["what"] = ("BASELINE configs[%d] per-GPU shape: %d users over %d GPU(s) x %d replicated dishes, E = %d; "
                   "per-shard top-%d for every user in rounds of %d + ONE all-gather of [shard, %d] x (f32, i32)%s; "
                   "max over ranks"
                   % (3 if E == 64 else 4, users_total, world, I, E, k, round_users, k,
                      "" if dist is not None else " (single process: no peers, no collective)"))
    eng.close()
    del PM, RE, CE, dish_cats, eng, sh
    if torch.device(dev).type == "cuda":
        torch.cuda.empty_cache()
    return out


def projected_world8_block(torch, foodrec_amd, dev, users_total, I, E, k, topk_path_ms_n1,
                           rounds=(262144, 0, 524288), repeats=3):
    """A ONE-GPU PROJECTION of the sharded top-k path at 8 GPUs -- not a measurement of 8 GPUs: this box has one.
    What one GPU can say: how long the N = 8 per-GPU shape takes (BASELINE configs[3]: users_total / 8 users held as
    the LAST shard of eight, the same replicated catalogue, ranked in rounds), at several round sizes.  What it cannot
    say is what the seven peers and the collective do; the exchange is priced from SURVEY.md section 5's link rate
    instead.  `rounds`: users per round; 0 = the shard cut into the fewest EVEN rounds of at most 524 288."""
    from foodrec_amd.sharding import UserShardedScorer, shard_range
    C, world, rank = 4, 8, 7
    base, count = shard_range(users_total, world, rank)
    per = -(-users_total // world)
    g, RE, CE, dish_cats = _replicated_tables(torch, dev, I, C, E)
    g.manual_seed(20260101 + 40 + rank)
    PM = torch.randn((count, C + 1, E), generator=g, device=dev) * E ** -0.5
    eng = foodrec_amd.ScoringEngine(PM, RE, CE, coef=0.99, device=dev, user_base=base)
    eng.set_dish_categories(dish_cats)
    sh = UserShardedScorer(eng, users_total, device=dev)
    # this process plays rank 7 of 8 (no collective is issued)
    sh.base, sh.count, sh.per, sh.rank, sh.world = base, count, per, rank, world
    clk = Clock(torch, dev)
    out_rounds = []
    link = XGMI_LINK_GBS * 1e6                                  # bytes per ms
    for R in rounds:
        R = int(R) if R else -(-count // -(-count // 524288))
        first = torch.arange(base, base + min(R, count), dtype=torch.int32, device=dev)
        sh.topk_local(k, first)
        sh.topk_local(k, first)                                 # tables, scratch at this round's size
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
        # all-gather of one round's pieces over xGMI: every rank sends its piece to 7 peers over 7 links at once
        # (direct, what a fully connected topology allows) or around a ring (7 steps of one piece per link)
        direct_ms, ring_ms = piece / link, 7 * piece / link
        exposed_direct, exposed_ring = last_piece / link, 7 * last_piece / link
        n1 = topk_path_ms_n1
        out_rounds.append({
            "round_users": R, "rounds": nround, "last_round_users": last_rows, "shard_ms": shard_ms,
            "shard_ms_all": walls, "ms_per_round_avg": shard_ms / nround,
            "allgather_bytes_per_rank_per_round": piece,
            "allgather_ms_per_round_at_153GBps_direct": direct_ms,
            "allgather_ms_per_round_at_153GBps_ring": ring_ms,
            "hidden_behind_next_round": bool(ring_ms < shard_ms / nround),
            "exposed_last_round_ms_direct": exposed_direct, "exposed_last_round_ms_ring": exposed_ring,
            "implied_speedup_upper_bound": (n1 / (shard_ms + exposed_direct)) if n1 else None,
            "implied_speedup_with_ring_exchange": (n1 / (shard_ms + exposed_ring)) if n1 else None})
    best = min(out_rounds, key=lambda r: r["shard_ms"])
    eng.close()
    del PM, RE, CE, dish_cats, eng, sh
    if torch.device(dev).type == "cuda":
        torch.cuda.empty_cache()
    return {"status": "PROJECTION from one GPU: UNMEASURED ON HARDWARE at N = 8",
            "what": ("the N = 8 per-GPU shape of the sharded top-k path timed on ONE GPU: %d of %d users held as shard "
                     "[%d, %d), %d replicated dishes, E = %d, top-%d for every user of the shard in rounds; the "
                     "exchange priced at %.0f GB/s per xGMI link (SURVEY.md section 5), not run"
                     % (count, users_total, base, base + count, I, E, k, XGMI_LINK_GBS)),
            "topk_path_ms_n1": topk_path_ms_n1, "users_per_gpu": count, "rounds": out_rounds,
            "best_round_users": best["round_users"], "shard_ms": best["shard_ms"],
            "implied_speedup_upper_bound": best["implied_speedup_upper_bound"],
            "implied_speedup_with_ring_exchange": best["implied_speedup_with_ring_exchange"],
            "upper_bound_because": ("every rank is assumed as fast as this one, the per-round collectives fully hidden "
                                    "behind the next round's ranking (they take a few per cent of a round at the link "
                                    "rate), launch and host overheads as on this box; north_star asks for >= 6x"),
            "north_star_target": 6.0}


def routed_pairs_leg(torch, dist, eng, U, I, C, dev, world, B, repeats=5):
    """Every rank brings B pairs whose users are spread over ALL shards; UserShardedScorer.score_pairs_routed buckets
    them by owner, all-to-alls the records, the owners score, the scores come back ('pairs routed to
    the owner of the user').  Whole-job pairs/s over the median max-over-ranks wall time."""
    from foodrec_amd.sharding import UserShardedScorer
    sh = UserShardedScorer(eng, world * U, device=dev, always_collective=True)
    g = torch.Generator(device=dev)
    g.manual_seed(900 + dist.get_rank())
    users = torch.randint(0, world * U, (B,), generator=g, device=dev, dtype=torch.int32)
    items = torch.randint(0, I, (B,), generator=g, device=dev, dtype=torch.int32)
    _, cats = random_masks(torch, B, C, dev, g)
    sh.score_pairs_routed(users, items, cats)             # warm-up, with the collective id check
    walls = []
    clk = Clock(torch, dev)
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
    # the pairs this rank owns: same kernel, same bits
    idx = (sh.owner_of(users) == dist.get_rank()).nonzero(as_tuple=True)[0]
    ok = bool(idx.numel() == 0 or torch.equal(eng.score_pairs(users[idx], items[idx], cats[idx]), out[idx]))
    eng.check()
    wall = median(walls)
    return {"pairs_per_gpu": B, "repeats": repeats, "wall_ms_median": wall * 1e3,
            "wall_ms_all": [w * 1e3 for w in walls], "pairs_per_s_whole_job": world * B / wall,
            "bytes_per_pair_on_the_wire": (2 + C) * 4 + 4, "own_pairs_match_local_scoring": ok,
            "what": "bucket by owner (one device sort) + all-to-all of (user, dish, mask) records + owner-side "
                    "m2d_score_pairs + all-to-all of f32 scores; includes the one host round trip for bucket sizes"}
