#!/usr/bin/env python3
"""profiles/traffic.json from a scripts/profile_gpu.sh run: the fabric bytes per launch of the timed pair kernel (PMC
passes) keyed by the workload's shape, with the streaming-read probe of the SAME box beside them (bench.py publishes the
reading only on a box whose probe is within 10 % of it).
    python scripts/update_traffic.py gpurun_out/prof_<tag> <a default bench.py line of the same gpurun call> [profiles/traffic.json]"""
import json
import sys

prof, bench_path = sys.argv[1], sys.argv[2]
dst = sys.argv[3] if len(sys.argv) > 3 else "profiles/traffic.json"
summ = json.load(open(prof + "/summary.json"))
line = json.loads(open(bench_path).read().strip().splitlines()[-1])
cfg = summ["bench_line"]["config"]
probe = line["roofline"]["stream_read_probe"]["GBps"]
name = max((k for k in summ["counters"] if "m2d_score_pairs" in k), key=lambda k: summ["kernels"].get(k, {}).get("total_ms", 0))
cs = summ["counters"][name]
read, write = cs["fabric_read_bytes_gfx950_corrected"], cs["fabric_write_bytes"]
skip = cfg["options"]["skip_masked"] != 0
key = "E%d_B%d_U%d_I%d%s" % (cfg["embed_size"], cfg["pairs_per_step_per_gpu"], cfg["users_per_gpu"], cfg["dishes"], "_skip" if skip else "")
try:
    tj = json.load(open(dst))
except Exception:
    tj = {}
tj[key] = {"fabric_bytes_per_launch": read + write, "read_bytes_corrected": read, "write_bytes": write,
           "stream_probe_GBps": probe, "kernel": name, "l2_hit_rate": cs.get("l2_hit_rate"),
           "source": "%s/summary.json (scripts/profile_gpu.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE of `bench.py --steps 50 "
                     "--warmup 5 --no-cpu-baseline --no-side`, separate passes; FETCH_SIZE x 1024 x 2 per MI355X_MICROARCH.md, HBM "
                     "section); stream probe from a default bench.py run of the same gpurun call" % prof.replace("gpurun_out/prof_", "profiles/"),
           "note": "FETCH_SIZE / WRITE_SIZE count L2<->fabric requests, Infinity-Cache hits included: fabric bytes, an upper bound of true HBM bytes"}
json.dump(tj, open(dst, "w"), indent=1)
print(key, tj[key]["fabric_bytes_per_launch"], probe)
