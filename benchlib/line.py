"""The nested legs of a line as top-level scalars: a record that keeps nested objects by key name only (the driver's
`parsed`) still carries one number per BASELINE config and per roofline figure."""
from __future__ import annotations


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def config_scalars(line):
    """cfgN_* = BASELINE.json configs[N]; the rest are SURVEY.md 8d's figures of the default (configs[1]) run."""
    roof = line.get("roofline") or {}
    cfg1 = str((line.get("config") or {}).get("workload", "")).lstrip("[").startswith("BASELINE configs[1]:") \
        and line.get("n_gpus") == 1
    out = {
        "cfg0_evaluator_users_per_s": _get(line, "evaluator", "device_users_per_s_second_call"),
        "cfg1_pairs_per_s": line.get("value") if cfg1 else None,
        "cfg1_hbm_frac": roof.get("frac") if cfg1 and roof.get("bound") == "hbm" else None,
        "cfg1_ingredients_pairs_per_s": _get(line, "with_ingredient_table", "pairs_per_s"),
        "cfg1_ingredients_hbm_frac": _get(line, "with_ingredient_table", "frac"),
        "cfg2_mlp_e128_pairs_per_s": _get(line, "config2_mlp", "pairs_per_s"),
        "cfg2_mlp_e128_ms": _get(line, "config2_mlp", "kernel_avg_ms"),
        "cfg2_mlp_mfma_frac": _get(line, "config2_mlp", "roofline", "frac"),
        "cfg2_mlp_max_rel_vs_restatement": _get(line, "config2_mlp", "max_rel_vs_restatement"),
        "cfg3_topk_e64_path_ms": _get(line, "scaling_path", "wall_ms"),
        "cfg3_topk_e64_mfma_frac": _get(line, "scaling_path", "roofline_frac_of_mfma_peak"),
        "cfg4_topk_e128_round_ms": _get(line, "config4_topk", "round_ms"),
        "cfg4_topk_e128_round_users": _get(line, "config4_topk", "round_users"),
        "cfg4_topk_e128_mfma_frac": _get(line, "config4_topk", "roofline", "frac"),
        "survey_8d_pairs_per_s": roof.get("survey_8d_pairs_per_s"),
        "survey_8d_frac": roof.get("survey_8d_frac"),
        "hbm_only_frac_of_spec": roof.get("hbm_only_frac_of_spec"),
        "hbm_only_masked_frac_of_spec": roof.get("hbm_only_masked_frac_of_spec"),
        "hbm_only_frac_of_stream_probe": roof.get("hbm_only_frac_of_stream_probe"),
        "stream_probe_GBps": roof.get("stream_probe_GBps"),
        "topk_pruned_frac": _get(line, "catalogue_topk", "roofline", "frac"),
        "topk_every_tile_frac": _get(line, "catalogue_topk", "every_tile", "roofline", "frac"),
        "cpu_baseline_pairs_per_s": _get(line, "cpu_baseline", "value"),
        "parity": "partial: oracle arithmetic unpinned (TensorFlow not importable, reference ships no fixtures)",
    }
    return {k: v for k, v in out.items() if v is not None}


def scaling_scalars(scaling):
    """`scaling_path` as scalars of the line itself: the path north_star's '>= 6x at 8 GPUs' speaks of."""
    out = {"topk_path_ms": scaling.get("wall_ms"),
           "topk_path_pairs_decided_per_s": scaling.get("pairs_decided_per_s_whole_job"),
           "topk_path_pairs_multiplied_per_s": scaling.get("pairs_multiplied_per_s_whole_job"),
           "topk_path_allgather_exposed_ms": scaling.get("allgather_exposed_ms"),
           "topk_path_allgather_exposed_ms_max": scaling.get("allgather_exposed_ms_max"),
           "topk_path_allgather_exposed_ms_min": scaling.get("allgather_exposed_ms_min"),
           "topk_path_dtype": "bf16x3" if str(scaling.get("kernel", "")).endswith("bf16x3") else "f32",
           "topk_path_users_total": scaling.get("users_total"), "topk_path_dishes": scaling.get("dishes")}
    pj = scaling.get("projected_world8") or {}
    if "shard_ms" in pj:                                        # (one-GPU projection of N = 8: labelled as such)
        out.update({"topk_path_projected_world8_shard_ms": pj["shard_ms"],
                    "topk_path_projected_world8_speedup_upper_bound": pj["implied_speedup_upper_bound"],
                    "topk_path_projected_world8_status": pj["status"]})
    return out


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data")
ROOFLINE_FIRST = ("bound", "achieved", "peak", "unit", "frac", "traffic", "survey_8d_frac", "survey_8d_pairs_per_s",
                  "hbm_only_frac_of_spec", "hbm_only_masked_frac_of_spec", "stream_probe_GBps",
                  "hbm_only_frac_of_stream_probe", "hbm_only_masked_frac_of_stream_probe", "kernel_avg_ms",
                  "algorithmic_bytes_per_pair", "pairs_per_launch")
# the scalars a record that keeps only the END of the line (the driver's `tail`: 2 000 characters) must still show;
# last = most wanted
TAIL_KEYS = ("ranks_seen", "distinct_devices", "dist_backend", "topk_path_ms", "topk_path_allgather_exposed_ms_max",
             "topk_path_projected_world8_speedup_upper_bound", "cfg0_evaluator_users_per_s", "cfg1_pairs_per_s",
             "cfg1_hbm_frac", "cfg1_ingredients_pairs_per_s", "cfg3_topk_e64_path_ms", "cfg3_topk_e64_mfma_frac",
             "topk_pruned_frac", "topk_every_tile_frac", "cpu_baseline_pairs_per_s", "stream_probe_GBps",
             "hbm_only_frac_of_spec", "hbm_only_masked_frac_of_spec", "survey_8d_pairs_per_s", "survey_8d_frac",
             "cfg4_topk_e128_round_ms", "cfg4_topk_e128_mfma_frac", "cfg2_mlp_e128_ms",
             "cfg2_mlp_max_rel_vs_restatement", "cfg2_mlp_mfma_frac", "cfg2_mlp_e128_pairs_per_s", "parity")


def _short(v):
    """Six significant digits for the tail's copies (the nested legs keep every digit)."""
    return float("%.6g" % v) if isinstance(v, float) else v


def ordered(line):
    """The same line, keys in the order a truncating reader serves best: the contract's keys first; `roofline` with its
    headline scalars ahead of the nested legs; the nested legs in the middle; every other top-level scalar at the END,
    the per-config summary last (the driver's record keeps the last 2 000 characters of stdout verbatim)."""
    out = {k: line[k] for k in CONTRACT_KEYS if k in line}
    if "config" in line:
        out["config"] = line["config"]
    roof = line.get("roofline")
    if isinstance(roof, dict):
        r2 = {k: roof[k] for k in ROOFLINE_FIRST if k in roof}
        r2.update({k: v for k, v in roof.items() if k not in r2 and not isinstance(v, (dict, list))})
        r2.update({k: v for k, v in roof.items() if k not in r2})
        out["roofline"] = r2
    if "cpu_baseline" in line:
        out["cpu_baseline"] = line["cpu_baseline"]
    out.update({k: v for k, v in line.items() if k not in out and isinstance(v, (dict, list))})
    out.update({k: v for k, v in line.items() if k not in out and k not in TAIL_KEYS})
    out.update({k: _short(line[k]) for k in TAIL_KEYS if k in line})
    return out
