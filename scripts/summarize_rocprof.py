#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + per-counter PMC passes) into one JSON/markdown summary.

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for
wide coalesced reads (MI355X_MICROARCH.md, HBM section), so the read side is doubled before it is
compared with a byte count.  The raw value is kept beside the corrected one.  Both counters sit on the L2's
memory-side (fabric) port: Infinity-Cache hits are counted, so the numbers are named fabric bytes, not HBM bytes."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def rows(pattern):
    for path in glob.glob(pattern, recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                yield r


def main():
    out, tag = sys.argv[1], sys.argv[2]
    summ = {"tag": tag, "kernels": {}, "counters": {}}
    # kernel trace: per-dispatch durations
    dur = defaultdict(list)
    for r in rows(os.path.join(out, "stats", "**", "*kernel_trace.csv")):
        name = r.get("Kernel_Name", "")
        dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    order = defaultdict(list)
    for r in rows(os.path.join(out, "stats", "**", "*kernel_trace.csv")):
        order[r.get("Kernel_Name", "")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for name, d in dur.items():
        last = [x[1] for x in sorted(order[name])[-10:]]          # the warm launches: the last 10 dispatches of the kernel
        summ["kernels"][name] = {"calls": len(d), "avg_us": sum(d) / len(d) / 1e3, "min_us": min(d) / 1e3,
                                 "max_us": max(d) / 1e3, "total_ms": sum(d) / 1e6, "last10_avg_us": sum(last) / len(last) / 1e3}
    # counters: average per dispatch per kernel
    for d in glob.glob(os.path.join(out, "pmc_*")):
        if not os.path.isdir(d):
            continue
        acc = defaultdict(lambda: defaultdict(list))
        for r in rows(os.path.join(d, "**", "*counter_collection.csv")):
            acc[r.get("Kernel_Name", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for kname, cs in acc.items():
            for cname, vals in cs.items():
                summ["counters"].setdefault(kname, {})[cname] = {"avg_per_dispatch": sum(vals) / len(vals), "dispatches": len(vals)}
    for kname, cs in summ["counters"].items():
        if "FETCH_SIZE" in cs:
            raw = cs["FETCH_SIZE"]["avg_per_dispatch"] * 1024
            # L2 -> fabric read requests (TCC_EA0_RDREQ): Infinity-Cache hits are INCLUDED, so these are fabric bytes,
            # an upper bound on what actually came from HBM
            cs["fabric_read_bytes_raw"] = raw
            cs["fabric_read_bytes_gfx950_corrected"] = raw * 2
        if "WRITE_SIZE" in cs:
            cs["fabric_write_bytes"] = cs["WRITE_SIZE"]["avg_per_dispatch"] * 1024
        if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs:
            h, m = cs["TCC_HIT_sum"]["avg_per_dispatch"], cs["TCC_MISS_sum"]["avg_per_dispatch"]
            cs["l2_hit_rate"] = h / (h + m) if h + m else None
    try:
        summ["bench_line"] = json.loads(open(os.path.join(out, "stats_bench.json")).read().strip().splitlines()[-1])
    except Exception as e:                                             # noqa: BLE001
        summ["bench_line"] = "unavailable: %s" % e
    # MFMA profile (scripts/prof_mfma.py): the roofline fractions of its JSON, recomputed from the kernel trace
    try:
        mb = json.load(open(os.path.join(out, "mfma_bench.json")))
        summ["mfma_bench"] = mb
        summ["mfma_fractions_from_trace"] = {}
        for key, v in mb.items():
            base = {"m2d_topk_grouped_bf16x3": "m2d_topk_grouped_bf16_pipe2", "m2d_topk_grouped": "m2d_topk_grouped<",
                    "m2d_mlp_pc_bf16x3": "m2d_mlp_pc<"}.get(v["kernel"], v["kernel"])
            cands = [(n, k) for n, k in summ["kernels"].items() if base in n and "merge" not in n]
            if v["kernel"] == "m2d_topk_grouped_bf16x3" and "block_users" in v:      # blocks of 128 users: the four-wave instantiation
                tail = ", 4, " if v["block_users"] == 128 else ", 8, "       # ...<E, KR, 1, false, WAVES, KEEP>
                cands = [(n, k) for n, k in cands if tail in n] or cands
            if not cands:
                continue
            n, k = max(cands, key=lambda nk: nk[1]["total_ms"])
            avg_us = k["last10_avg_us"]
            if "scan_kernel_dispatches" in v:                # the leg's own timed launches of that kernel
                a0, a1 = v["scan_kernel_dispatches"]
                mine = [x[1] for x in sorted(order[n])[a0:a1]]
                if mine:
                    avg_us = sum(mine) / len(mine) / 1e3
            frac = v["executed_flop_per_launch"] / (avg_us * 1e-6) / 1e12 / v["peak_TFLOPs"]
            summ["mfma_fractions_from_trace"][key] = {"trace_kernel": n, "trace_last10_avg_us": avg_us,
                                                      "event_avg_ms": v["event_avg_ms"], "frac_from_trace": frac,
                                                      "frac_from_events": v["frac_of_peak"]}
    except Exception:                                                  # noqa: BLE001
        pass
    with open(os.path.join(out, "summary.json"), "w") as f:
        json.dump(summ, f, indent=1)
    with open(os.path.join(out, "summary.md"), "w") as f:
        f.write("# rocprofv3 summary %s\n\n| kernel | calls | avg us | min us | max us | avg of last 10 us |\n|---|---|---|---|---|---|\n" % tag)
        for k, v in sorted(summ["kernels"].items(), key=lambda kv: -kv[1]["total_ms"]):
            f.write("| %s | %d | %.1f | %.1f | %.1f | %.1f |\n" % (k[:90], v["calls"], v["avg_us"], v["min_us"], v["max_us"], v["last10_avg_us"]))
        if summ.get("mfma_fractions_from_trace"):
            f.write("\n## roofline fractions, recomputed from the trace (executed flop per launch / last-10 average / peak)\n\n"
                    "| leg | kernel | trace avg us (the leg's timed launches) | HIP-event avg ms (whole call) | frac from trace | frac from events |\n|---|---|---|---|---|---|\n")
            for key, v in summ["mfma_fractions_from_trace"].items():
                f.write("| %s | %s | %.1f | %.3f | %.3f | %.3f |\n" % (key, v["trace_kernel"][:60], v["trace_last10_avg_us"], v["event_avg_ms"],
                                                                    v["frac_from_trace"], v["frac_from_events"]))
        f.write("\n## counters (average per dispatch)\n\n")
        for k, cs in summ["counters"].items():
            f.write("### %s\n\n" % k[:120])
            for c, v in cs.items():
                f.write("- %s: %s\n" % (c, v["avg_per_dispatch"] if isinstance(v, dict) else v))
            f.write("\n")
    print(open(os.path.join(out, "summary.md")).read())


if __name__ == "__main__":
    main()
