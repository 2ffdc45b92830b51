"""Full-catalogue top-k (m2d_topk_users): the side leg of the default run and the MFMA roofline of a timed top-k
step.  The flops priced are the flops EXECUTED: the pattern-grouped kernels step through the tiles of the mask
patterns that can reach a block's top-k only."""
from __future__ import annotations

import time

from .common import BF16_MFMA_PEAK_TFLOPS, F32_MFMA_PEAK_TFLOPS, bare_loop_fields, median, random_masks

X3_ALL = "split bf16 (x = hi + lo, 3 x v_mfma_f32_32x32x16_bf16, fp32 accumulate)"
X3_HI_FIRST = ("split bf16 (x = hi + lo): hi x hi for every tile, lo x hi + hi x lo for the tiles that can hold a "
               "candidate (v_mfma_f32_32x32x16_bf16, fp32 accumulate)")
F32_EXACT = "f32 (v_mfma_f32_32x32x2_f32, exact)"
NOTE_X3 = ("pipelined kernel; `frac` prices the flops EXECUTED: users are sorted by the mask patterns that can reach "
           "their top-k and a block steps through those patterns' tiles only (Cauchy-Schwarz bounds widened by the "
           "rounding of the sums of absolute terms; DESIGN.md 4.4), so most (user, dish) pairs are decided without "
           "being multiplied -- pairs_per_s counts every pair of the catalogue")


def completed_count(eng):
    """(wave, tile) pairs given the two cross products by the hi x hi first form; -1 when the form did not run."""
    return eng.get_option("topk_tiles_completed")


def catalogue_topk_leg(torch, eng, U, I, C, E, dev, user_base, n_users, k=10, keep=None, reps=7):
    """Full-catalogue top-k for n_users users.  `keep`: a dict that receives the last call's lists."""
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    _, dish_cats = random_masks(torch, I, C, dev, g)
    eng.set_dish_categories(dish_cats)
    users = (torch.randperm(U, generator=g, device=dev)[:n_users].to(torch.int32) + int(user_base)).contiguous()
    eng.topk_users(users[:1024], k)                       # builds the retrieval tables
    t_warm = time.perf_counter()                          # the first full launches run 5-10 % slow (clock ramp): at
    for i in range(40):                                   # least three, and 60 ms of them
        eng.topk_users(users, k)
        if i >= 2:
            torch.cuda.synchronize()
            if time.perf_counter() - t_warm > 0.06:
                break
    torch.cuda.synchronize()
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
    grouped = kernel.startswith("m2d_topk_grouped")
    dense = 2.0 * (C + 1) * E * n_users * I                 # the [users x (C+1)E] . [(C+1)E x dishes] contraction
    flops = 2.0 * E * n_users * I if grouped else dense     # the pattern-grouped kernel (0/1 masks) contracts over E
    x3 = kernel.endswith("bf16x3")                         # 3 bf16 MFMAs per 16 k-values
    scanned = full = None
    if grouped:
        scanned, full = eng.get_option("topk_tiles_scanned"), eng.get_option("topk_tiles_full")
        if scanned > 0:                                     # a block's user lanes (256, or 128) x a tile's 32 dishes
            flops = 2.0 * E * eng.get_option("topk_block_users") * 32 * scanned
    # split bf16: three products per tile -- or, the hi x hi first form (large catalogues), one product per tile and the
    # two cross products for the (wave, tile) pairs that could still hold a candidate
    completed = completed_count(eng) if grouped else -1
    ex3 = 3 * flops if completed < 0 else flops + 2 * (2.0 * E * 32 * 32 * completed)
    frac_scanned = scanned / full if scanned and full else None
    common = {"tiles_scanned": scanned, "tiles_without_pruning": full, "scanned_fraction": frac_scanned}
    if x3:
        tf = ex3 / ms / 1e9
        roof = {"bound": "mfma", "achieved": tf, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / BF16_MFMA_PEAK_TFLOPS, "flop_per_pair": ex3 / n_users / I,
                "hi_first_form": completed >= 0,
                "wave_tiles_given_cross_products": completed if completed >= 0 else None,
                "frac_if_priced_as_three_products": (3 * flops / ms / 1e9 / BF16_MFMA_PEAK_TFLOPS
                                                     if completed >= 0 else None),
                **common,
                "frac_if_every_tile_were_scanned": 3 * 2.0 * E * n_users * I / ms / 1e9 / BF16_MFMA_PEAK_TFLOPS,
                "dtype": X3_HI_FIRST if completed >= 0 else X3_ALL, **bare_loop_fields(tf), "note": NOTE_X3}
    else:
        tf = flops / ms / 1e9
        roof = {"bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / F32_MFMA_PEAK_TFLOPS, "dtype": F32_EXACT, "flop_per_pair": flops / n_users / I, **common,
                "frac_if_every_tile_were_scanned": 2.0 * E * n_users * I / ms / 1e9 / F32_MFMA_PEAK_TFLOPS,
                "note": "`frac` prices the flops EXECUTED (tiles of the blocks' relevant patterns)"}
    return {"users": n_users, "dishes": I, "k": k, "median_ms": ms, "users_per_s": n_users / ms * 1e3,
            "pairs_per_s": n_users * I / ms * 1e3, "tflops": flops / ms / 1e9,
            "dense_equivalent_tflops": dense / ms / 1e9, "roofline": roof, "kernel": kernel}


def compare_lists(torch, a, b):
    """Dish ids are index output: how the default (split-bf16) lists differ from the exact-f32 kernel's for the same
    users.  Both kernels finish near-tied lists in one arithmetic (m2d_topk_refine; option topk_refine), so they
    should not; `max_gap_at_mismatch` is the largest |score difference| between the two kernels at a position that
    holds different dishes, relative to max(1, |score|)."""
    ia, ib, sa, sb = a["ids"], b["ids"], a["scores"], b["scores"]
    diff = ia != ib
    rows = diff.any(dim=1)
    rel = (sa - sb).abs() / sb.abs().clamp(min=1.0)
    gap = rel[diff]
    return {"lists_identical_frac": 1.0 - float(rows.float().mean().item()),
            "lists_differing": int(rows.sum().item()), "positions_differing": int(diff.sum().item()),
            "max_gap_at_mismatch": float(gap.max().item()) if gap.numel() else 0.0,
            "max_score_difference": float(rel.nan_to_num(nan=0.0).max().item()),
            "what": "default split-bf16 lists against the exact-f32 kernel's (option topk_bf16x3 = 0), same users"}


class _Opt:
    """`with _Opt(eng, name, value):` -- an engine option for the length of a block."""

    def __init__(self, eng, name, value):
        self.eng, self.name, self.value = eng, name, value

    def __enter__(self):
        self.was = self.eng.get_option(self.name)
        self.eng.set_option(self.name, self.value)

    def __exit__(self, *exc):
        self.eng.set_option(self.name, self.was)


def catalogue_topk_block(torch, eng, U, I, C, E, dev, user_base, n_users):
    """`catalogue_topk` of the default line: the split-bf16 kernel, the exact-f32 kernel on the same users, how their
    lists differ, what the refinement of near-tied lists costs, and the same kernel made to step through every tile
    (topk_prune = 0: the MFMA-bound form)."""
    args = (torch, eng, U, I, C, E, dev, user_base, n_users)
    lists_x3, lists_f32 = {}, {}
    blk = catalogue_topk_leg(*args, keep=lists_x3)
    with _Opt(eng, "topk_bf16x3", 0):                    # the exact-f32 kernel's figure beside the split-bf16 one
        blk["exact_f32"] = catalogue_topk_leg(*args, keep=lists_f32)
    blk["index_exactness"] = compare_lists(torch, lists_x3, lists_f32)
    blk["index_exactness"].update({"refined_users": eng.get_option("topk_refined"),
                                   "refined_users_sent_to_the_repair": eng.get_option("topk_refine_repaired")})
    del lists_x3, lists_f32
    with _Opt(eng, "topk_refine", 0):                    # what finishing the near-tied lists in one arithmetic costs
        lists_off, lists_f32_off = {}, {}
        off = catalogue_topk_leg(*args, keep=lists_off)
        blk["refine_off"] = {"median_ms": off["median_ms"], "pairs_per_s": off["pairs_per_s"],
                             "what": "option topk_refine = 0: the split-bf16 lists as the scan leaves them; ids then "
                                     "differ from the exact-f32 kernel's wherever two scores sit inside the split's "
                                     "rounding"}
        with _Opt(eng, "topk_prune", 0):                 # the every-tile form without it: the scan's own fraction
            et = catalogue_topk_leg(*args)
            blk["refine_off"]["every_tile"] = {"median_ms": et["median_ms"], "roofline_frac": et["roofline"]["frac"]}
        with _Opt(eng, "topk_bf16x3", 0):
            catalogue_topk_leg(*args, keep=lists_f32_off)
        blk["refine_off"]["index_exactness"] = compare_lists(torch, lists_off, lists_f32_off)
        del lists_off, lists_f32_off
    with _Opt(eng, "topk_prune", 0):                     # the same kernel made to step through every tile
        blk["every_tile"] = catalogue_topk_leg(*args)
    return blk


def timed_topk_roofline(eng, kernel_used, C, E, units, avg_ms, with_ingredients):
    """MFMA roofline of a timed top-k step (--workload topk / --config 3|4) of `units` (user, dish) pairs.
    Returns (roofline dict, scanned fraction or None, dtype label)."""
    K = (C + 1) * E
    x3 = kernel_used.endswith("bf16x3")
    grouped = kernel_used.startswith("m2d_topk_grouped")
    Ew = 2 * E if with_ingredients else E                 # grouped rows are [H[d] | RE[d]] with the ingredient table
    fl = (2.0 * Ew * (3 if x3 else 1) if grouped else 2.0 * K) * units
    # executed flops = the scanned share of the catalogue's (the step's last launch stands for the step)
    scanned_frac = None
    if grouped and not with_ingredients:
        sc_, fu_ = eng.get_option("topk_tiles_scanned"), eng.get_option("topk_tiles_full")
        if sc_ > 0 and fu_ > 0:
            scanned_frac = sc_ / fu_
    tf_all = fl / (avg_ms * 1e-3) / 1e12
    tf = tf_all * (scanned_frac if scanned_frac is not None else 1.0)
    # the hi x hi first form: one product per tile stepped through, the two cross products for the share of (wave,
    # tile) pairs that could still hold a candidate
    cross_share = None
    if x3 and grouped:
        cmp_ = completed_count(eng)
        sc2_ = eng.get_option("topk_tiles_scanned")      # (with the ingredient table: every tile, no pattern pruned)
        if cmp_ >= 0 and sc2_ > 0:
            cross_share = cmp_ * 32.0 / (sc2_ * eng.get_option("topk_block_users"))
            tf = tf / 3.0 * (1.0 + 2.0 * cross_share)
    peak = BF16_MFMA_PEAK_TFLOPS if x3 else F32_MFMA_PEAK_TFLOPS
    per_pair = fl / units if cross_share is None else fl / units / 3.0 * (1.0 + 2.0 * cross_share)
    roof = {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "traffic": None,
            "step_avg_ms": avg_ms, "flop_per_pair_executed": per_pair,
            "dense_equivalent_tflops": 2.0 * K * units / (avg_ms * 1e-3) / 1e12,
            "scanned_fraction": scanned_frac, "frac_if_every_tile_were_scanned": tf_all / peak,
            "hi_first_form": cross_share is not None, "share_of_wave_tiles_given_cross_products": cross_share,
            # the same tiles priced as the three-product form would execute them
            "frac_if_priced_as_three_products": (tf * 3.0 / (1.0 + 2.0 * cross_share) / peak
                                                 if cross_share is not None else None),
            "note": "`frac` prices the flops EXECUTED (the tiles the blocks stepped through), and `value` counts the "
                    "pairs of those tiles; pairs_decided_per_s counts every (user, dish) pair of the catalogue -- most "
                    "are decided by a bound, without being multiplied",
            "dtype": ((X3_HI_FIRST if cross_share is not None else X3_ALL) if x3 else F32_EXACT)}
    if x3:
        roof.update(bare_loop_fields(tf))
    return roof, scanned_frac, ("bf16x3" if x3 else "f32")
