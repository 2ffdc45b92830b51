"""Side legs of the pair path (outside the timed region): SURVEY.md 8d's count on a run that fetches every row, the
no-reuse batch, the streaming-read probe, the HBM-only estimate, and the two table options."""
from __future__ import annotations

from .common import (HBM_PEAK_GBS, INFINITY_CACHE_BYTES, algorithmic_bytes_per_pair, median, random_masks, settle,
                     time_steps)

CACHE_LABEL = "includes cache-served bytes"


def survey_8d_leg(torch, eng, users, items, cats, C, E, steps):
    """SURVEY.md 8d's count charges every row of the user block (1 564 B per pair at E = 64).  The timed kernel leaves
    out the rows of weight-0 categories, so that count cannot be applied to ITS time: the same batch is timed once
    more with every row fetched (option skip_masked = 0, into a buffer of its own) and the survey's formula is
    published from that time."""
    was = eng.get_option("skip_masked")
    so = torch.empty(users.numel(), dtype=torch.float32, device=users.device)
    eng.set_option("skip_masked", 0)
    try:
        time_steps(torch, eng, users, items, cats, so, 3)
        _, per = time_steps(torch, eng, users, items, cats, so, max(10, min(steps, 50)))
        eng.check()
        kernel = eng.last_kernel()
    finally:
        eng.set_option("skip_masked", was)
    ms = sum(per) / len(per)
    B, bpp = users.numel(), algorithmic_bytes_per_pair(C, E)
    gbps = bpp * B / (ms * 1e-3) / 1e9
    return {"survey_8d_ms": ms, "survey_8d_bytes_per_pair": bpp, "survey_8d_GBps": gbps,
            "survey_8d_frac": gbps / HBM_PEAK_GBS, "survey_8d_pairs_per_s": B / (ms * 1e-3),
            "survey_8d_kernel": kernel + " (option skip_masked = 0: every row of the user block fetched)"}


def hbm_only_estimate(pairs, C, E, dishes, active_rows, kernel_ms, probe_gbps):
    """Bytes of a no-reuse launch that cannot come from a cache, over the kernel's time.

    Personal_Memory blocks are read once per launch from a table far larger than the Infinity Cache (non-temporal
    loads), the id / mask / score streams are touched once: those are HBM bytes.  The dish rows are counted as HBM
    bytes only when the dish table CANNOT be cache-resident (dishes x E x 4 above the cache's capacity); a table that
    fits is ASSUMED cache-served and left out.  A figure above this box's streaming-read probe cannot be DRAM-only:
    it is then published under the label `includes cache-served bytes` and no `frac_of_stream_probe` is given."""
    pm_bytes = pairs * (1.0 + active_rows) * E * 4
    stream_bytes = pairs * (C * 4 + 12)
    dish_table = dishes * E * 4
    dish_cached = dish_table <= INFINITY_CACHE_BYTES
    nbytes = pm_bytes + stream_bytes + (0 if dish_cached else pairs * E * 4)
    gbps = nbytes / kernel_ms / 1e6
    over = probe_gbps is not None and gbps > 1.02 * probe_gbps
    out = {"achieved": gbps, "unit": "GB/s", "frac_of_spec_peak": gbps / HBM_PEAK_GBS,
           "frac_of_stream_probe": None if (over or not probe_gbps) else gbps / probe_gbps,
           "bytes_per_launch": nbytes, "kernel_median_ms": kernel_ms,
           "dish_rows": ("assumed cache-served, left out: the %.0f MB dish table fits the %d MiB Infinity Cache"
                         % (dish_table / 1e6, INFINITY_CACHE_BYTES >> 20)) if dish_cached else
                        ("counted: the %.0f MB dish table exceeds the %d MiB Infinity Cache (part of it is still "
                         "served on-die)" % (dish_table / 1e6, INFINITY_CACHE_BYTES >> 20)),
           "label": CACHE_LABEL if over else "HBM bytes only"}
    if over:
        out["over_stream_probe_ratio"] = gbps / probe_gbps
    return out


def stream_probe_leg(torch, eng, buf):
    """m2d_stream_read_probe over `buf`: the achievable-peak figure SURVEY.md 8d asks for beside the spec peak."""
    sink = torch.zeros(4, device=buf.device)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(12)]
    nbytes = buf.numel() * 4
    for i in range(11):
        evs[i].record()
        eng.stream_read_probe(buf, sink)
    evs[11].record()
    torch.cuda.synchronize()
    ms = median([evs[i].elapsed_time(evs[i + 1]) for i in range(1, 11)])
    return {"bytes": nbytes, "median_ms": ms, "GBps": nbytes / ms / 1e6,
            "what": "m2d_stream_read_probe: plain streaming read of Personal_Memory, 16 B per lane, non-temporal"}


def side_measurements(torch, eng, PM, U, I, C, E, dev, user_base):
    """(i) the same kernel on a batch in which every user occurs at most once (no cache reuse of Personal_Memory rows
    at all), (ii) a plain streaming read of Personal_Memory, (iii) the HBM-only estimate from (i) -- all-ones masks
    and the benchmark's masks."""
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
    probe = stream_probe_leg(torch, eng, PM)
    hbm_only = hbm_only_estimate(Bn, C, E, I, float(C), ms, probe["GBps"])
    hbm_only["what"] = "no-reuse leg: Personal_Memory blocks + id / mask / score streams (+ dish rows, see dish_rows)"
    # the same with the benchmark's masks (random non-empty subsets): rows of absent categories are not fetched
    g2 = torch.Generator(device=dev)
    g2.manual_seed(8)
    _, cats2 = random_masks(torch, Bn, C, dev, g2)
    skip = eng.get_option("skip_masked") != 0
    act = float(cats2.sum(1).mean().item()) if skip else float(C)
    time_steps(torch, eng, users, items, cats2, out, 3)
    _, per2 = time_steps(torch, eng, users, items, cats2, out, 20)
    t2 = median(per2)
    masked = hbm_only_estimate(Bn, C, E, I, act, t2, probe["GBps"])
    masked.update({"mean_active_categories": act, "pairs_per_s": Bn / t2 * 1e3,
                   "what": "the same no-reuse batch with the benchmark's masks (uniform non-empty subsets): HBM bytes "
                           "= U_high + the rows of the active categories + streams"})
    hbm_only["masked"] = masked
    return nr, probe, hbm_only


def side_scalars(achieved, nr, probe, hbm_only):
    """The side measurements as scalars of `roofline` (a record that keeps only scalar keys keeps these)."""
    m = hbm_only["masked"]
    return {"algorithmic_over_stream_probe": achieved / probe["GBps"],
            "algorithmic_over_stream_probe_note": "algorithmic bytes count every gathered row, rows re-read from the "
                                                  "caches included: this ratio may exceed 1",
            "stream_probe_GBps": probe["GBps"], "no_reuse_GBps": nr["achieved"], "no_reuse_frac": nr["frac"],
            "hbm_only_GBps": hbm_only["achieved"], "hbm_only_frac_of_spec": hbm_only["frac_of_spec_peak"],
            "hbm_only_frac_of_stream_probe": hbm_only["frac_of_stream_probe"], "hbm_only_label": hbm_only["label"],
            "hbm_only_masked_GBps": m["achieved"], "hbm_only_masked_frac_of_spec": m["frac_of_spec_peak"],
            "hbm_only_masked_frac_of_stream_probe": m["frac_of_stream_probe"], "hbm_only_masked_label": m["label"]}


def user_high_leg(torch, eng, users, items, cats, C, E):
    """The same batch with the serving option "user_high_table" (the high-level sum from the derived table
    <U_high[u], CE_c>, 16 B per pair, instead of the gathered U_high row).  Not the headline: the table keeps part of
    the forward pass across launches."""
    out = torch.empty(users.numel(), dtype=torch.float32, device=users.device)
    eng.set_option("user_high_table", 1)
    try:
        settle(torch, lambda: eng.score_pairs(users, items, cats, out=out))     # (behind the evaluator's host loop)
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
            "what": "option user_high_table = 1: sum_c m_c <U_high[u], CE_c> / n from a 16 B-per-user derived table "
                    "instead of the gathered E x 4-byte U_high row; same scores within 1e-6"}


def set_synthetic_ingredients(torch, eng, I, E, R, dev, gen):
    """SURVEY.md 8d config 2's extension table: R rows, 1..20 ingredients per dish (build-chosen), uniform ids."""
    lens = torch.randint(1, 21, (I,), generator=gen, device=dev)
    off = torch.zeros(I + 1, dtype=torch.int32, device=dev)
    off[1:] = torch.cumsum(lens, 0).to(torch.int32)
    nnz = int(off[-1].item())
    eng.set_ingredients(torch.randn((R, E), generator=gen, device=dev) * E ** -0.5, off,
                        torch.randint(0, R, (nnz,), generator=gen, device=dev, dtype=torch.int32))


def ingredients_bytes_per_pair(C, E, mean_active):
    """U_high + H[d] + RE[d] + the low-level rows of the active categories."""
    return (3.0 + mean_active) * E * 4 + C * 4 + 12


def ingredients_leg(torch, eng, users, items, cats, I, C, E, dev, R):
    """The same batch with BASELINE configs[1]'s 10k-row ingredient table on the high-level path (build-defined
    extension; --workload ingredients makes it the timed step)."""
    g = torch.Generator(device=dev)
    g.manual_seed(20260101 + 3)
    set_synthetic_ingredients(torch, eng, I, E, R, dev, g)
    out = torch.empty(users.numel(), dtype=torch.float32, device=dev)      # its own buffer: never the timed kernel's

    def step():
        eng.score_pairs_ingredients(users, items, cats, out=out)
    settle(torch, step)
    time_steps(torch, eng, users, items, cats, out, 3, step)
    _, per = time_steps(torch, eng, users, items, cats, out, 10, step)
    eng.check()
    kern = eng.last_kernel()
    eng.clear_ingredients()
    ms = median(per)
    B = users.numel()
    active = float((cats != 0).sum(1).float().mean().item()) if eng.get_option("skip_masked") != 0 else float(C)
    bpp = ingredients_bytes_per_pair(C, E, active)
    return {"ingredient_rows": R, "ingredients_per_dish": "uniform 1..20", "kernel": kern, "kernel_median_ms": ms,
            "pairs_per_s": B / ms * 1e3, "algorithmic_bytes_per_pair": bpp, "achieved": bpp * B / ms / 1e6,
            "unit": "GB/s", "frac": bpp * B / ms / 1e6 / HBM_PEAK_GBS,
            "what": "same pairs, high-level path from the per-dish multi-hot ingredient sum H[d] (segment-sum hoisted "
                    "to a per-table kernel, DESIGN.md 8.1); no reference counterpart"}


def knob_sweep(torch, eng, users, items, cats, out, bpp, restore, log):
    """--sweep: the pair kernel's knobs, one line each on stderr."""
    so = torch.empty_like(out)
    B = users.numel()
    for pf in (1, 2, 4):
        for nt in (0, 1):
            for bpc in (2, 4, 8, 16):
                eng.set_option("prefetch", pf)
                eng.set_option("nt_loads", nt)
                eng.set_option("blocks_per_cu", bpc)
                time_steps(torch, eng, users, items, cats, so, 3)
                _, per = time_steps(torch, eng, users, items, cats, so, 10)
                ms = median(per)
                log("sweep pf=%d nt=%d blocks_per_cu=%2d: %.3f ms  %.2f Gpairs/s  %.0f GB/s"
                    % (pf, nt, bpc, ms, B / ms / 1e6, B * bpp / ms / 1e6))
    for k, v in restore.items():
        eng.set_option(k, v)
