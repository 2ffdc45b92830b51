"""CPU-side checks of bench.py's driver-facing contract: flags, defaults, the algorithmic byte model."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("m2d_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_follow_survey_8d():
    b = _bench()
    # (C+2)*E*4 + C*4 + 12: SURVEY.md section 8d
    assert [b.algorithmic_bytes_per_pair(4, E) for E in (32, 64, 128, 200)] == [796, 1564, 3100, 4828]
    # mask-aware count: U_high + the rows of the categories a dish has; all four categories = the survey's count
    assert b.algorithmic_bytes_per_pair(4, 64, 4.0) == 1564
    assert b.algorithmic_bytes_per_pair(4, 64, 1.0) == 3 * 256 + 28
    assert b.HBM_PEAK_GBS == 8000.0


def test_hbm_only_never_exceeds_the_stream_probe_unlabelled():
    """benchlib.pairs.hbm_only_estimate: dish rows count as HBM bytes only when the dish table cannot be cache-resident
    (capacity, not half of it), and a figure above the box's streaming-read probe carries the label `includes cache-served
    bytes` and no frac_of_stream_probe (round 5 published 1.105 x the probe as HBM-only)."""
    from benchlib import pairs
    from benchlib.common import INFINITY_CACHE_BYTES
    n, C, E = 1 << 20, 4, 64
    # 1 M dishes x 64 x 4 = 256 000 000 B <= 256 MiB: fits -> assumed cache-served, left out
    fits = pairs.hbm_only_estimate(n, C, E, 1_000_000, 4.0, 0.25, 5700.0)
    assert 1_000_000 * E * 4 <= INFINITY_CACHE_BYTES and fits["dish_rows"].startswith("assumed cache-served")
    assert fits["bytes_per_launch"] == n * (5 * E * 4 + C * 4 + 12)
    # 1 M dishes x 128 x 4 = 512 MB: cannot be resident -> counted
    big = pairs.hbm_only_estimate(n, C, 128, 1_000_000, 4.0, 0.6, 5700.0)
    assert big["dish_rows"].startswith("counted") and big["bytes_per_launch"] == n * (6 * 128 * 4 + C * 4 + 12)
    for est in (fits, big, pairs.hbm_only_estimate(n, C, 128, 1_000_000, 4.0, 0.5, 5700.0)):
        over = est["achieved"] > 1.02 * 5700.0
        assert (est["label"] == pairs.CACHE_LABEL) == over
        assert (est["frac_of_stream_probe"] is None) == over
        if not over:
            assert est["frac_of_stream_probe"] <= 1.02


def test_committed_bench_lines_of_this_round_parse_and_respect_the_probe():
    """profiles/r06_bench_*.json: each file is ONE JSON object (bench.py --out), no frac_of_stream_probe above 1.02
    anywhere in it, and config.workload within 200 characters."""
    import glob

    def walk(o, path=""):
        if isinstance(o, dict):
            for k, v in o.items():
                yield from walk(v, path + "/" + k)
        elif isinstance(o, list):
            for i, v in enumerate(o):
                yield from walk(v, path + "/%d" % i)
        else:
            yield path, o
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_bench_*.json"))):
        line = json.load(open(f))
        for path, v in walk(line):
            if path.endswith("frac_of_stream_probe") and v is not None:
                assert v <= 1.02, (f, path, v)
        assert len(line["config"]["workload"]) <= 200, (f, len(line["config"]["workload"]))


def test_top_level_scalars_of_the_default_line():
    """benchlib.line.config_scalars: one scalar per BASELINE config and roofline figure, taken from the nested legs."""
    from benchlib import line as linelib
    nested = {"value": 6.5e9, "n_gpus": 1, "config": {"workload": "BASELINE configs[1]: x"},
              "roofline": {"bound": "hbm", "frac": 0.88, "survey_8d_pairs_per_s": 4.6e9, "survey_8d_frac": 0.91,
                           "hbm_only_frac_of_spec": 0.73, "hbm_only_masked_frac_of_spec": 0.66, "stream_probe_GBps": 6100.0},
              "config2_mlp": {"pairs_per_s": 1.3e9, "kernel_avg_ms": 1.6, "roofline": {"frac": 0.35}, "max_rel_vs_restatement": 2e-6},
              "config4_topk": {"round_ms": 28.0, "round_users": 500000, "roofline": {"frac": 0.3}},
              "with_ingredient_table": {"pairs_per_s": 5.6e9, "frac": 0.94},
              "catalogue_topk": {"roofline": {"frac": 0.2}, "every_tile": {"roofline": {"frac": 0.46}}},
              "scaling_path": {"wall_ms": 326.0, "roofline_frac_of_mfma_peak": 0.43}}
    sc = linelib.config_scalars(nested)
    for key in ("cfg2_mlp_e128_pairs_per_s", "cfg2_mlp_e128_ms", "cfg2_mlp_mfma_frac", "cfg2_mlp_max_rel_vs_restatement",
                "cfg4_topk_e128_round_ms", "cfg4_topk_e128_mfma_frac", "cfg1_ingredients_pairs_per_s", "survey_8d_pairs_per_s",
                "survey_8d_frac", "hbm_only_frac_of_spec", "hbm_only_masked_frac_of_spec", "stream_probe_GBps",
                "topk_every_tile_frac", "cfg1_pairs_per_s", "cfg3_topk_e64_path_ms", "parity"):
        assert key in sc, key
        assert not isinstance(sc[key], (dict, list))
    assert sc["cfg2_mlp_mfma_frac"] == 0.35 and sc["topk_every_tile_frac"] == 0.46 and sc["parity"].startswith("partial")
    assert "cfg2_mlp_e128_ms" not in linelib.config_scalars({"value": 1.0, "n_gpus": 1, "config": {}, "roofline": {}})


def test_summary_scalars_end_the_line():
    """benchlib.line.ordered: the driver's record keeps `parsed` (contract keys, config, roofline, cpu_baseline; other keys by NAME
    only) and the last 2 000 characters of stdout -- so the per-config summary scalars are emitted LAST, and `roofline` leads with
    its headline scalars, nested legs behind them."""
    from benchlib import line as linelib
    path = os.path.join(ROOT, "profiles", "r06_bench_default.json")
    line = json.load(open(path))
    text = json.dumps(linelib.ordered(line))
    assert json.loads(text).keys() == line.keys()
    tail = text[-2000:]
    for key in ("cfg2_mlp_e128_pairs_per_s", "cfg2_mlp_e128_ms", "cfg2_mlp_mfma_frac", "cfg2_mlp_max_rel_vs_restatement",
                "cfg4_topk_e128_round_ms", "cfg4_topk_e128_mfma_frac", "cfg1_ingredients_pairs_per_s", "survey_8d_pairs_per_s",
                "survey_8d_frac", "hbm_only_frac_of_spec", "hbm_only_masked_frac_of_spec", "stream_probe_GBps",
                "topk_every_tile_frac", "topk_path_ms", "ranks_seen", "distinct_devices", "parity"):
        assert '"%s": ' % key in tail, key
    roof = list(linelib.ordered(line)["roofline"])
    assert roof[:5] == ["bound", "achieved", "peak", "unit", "frac"]
    first_nested = min(i for i, k in enumerate(roof) if isinstance(line["roofline"][k], (dict, list)))
    for key in ("survey_8d_frac", "hbm_only_frac_of_spec", "stream_probe_GBps"):
        assert roof.index(key) < first_nested


def test_bench_sources_stay_auditable():
    """bench.py and benchlib/: lines of at most 120 characters; bench.py itself under 500 lines."""
    import glob
    files = [os.path.join(ROOT, "bench.py")] + sorted(glob.glob(os.path.join(ROOT, "benchlib", "*.py")))
    for f in files:
        for i, l in enumerate(open(f).read().splitlines(), 1):
            assert len(l) <= 120, (f, i, len(l))
    assert len(open(files[0]).read().splitlines()) < 500


def test_flags_and_defaults(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.gpus, a.workload, a.users, a.dishes, a.embed, a.pairs) == (1, "pairs", 1_000_000, 100_000, 64, 1 << 22)
    assert a.steps > 0 and a.warmup > 0
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "3"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 7, 3)


def test_round_size_of_the_sharded_topk_path():
    """--round-users 0 (the default): a shard is ranked in the fewest EVEN rounds of at most 524 288 users -- BASELINE configs[3]'s
    10 M users: 20 rounds of 500 000 at N = 1, 3 of 416 667 at N = 8 (1.25 M per GPU) -- and an explicit value is taken as is."""
    b = _bench()
    assert b.default_round_users(10_000_000, 0) == 500_000
    assert b.default_round_users(1_250_000, 0) == 416_667
    assert b.default_round_users(2_500_000, 0) == 500_000 and b.default_round_users(5_000_000, 0) == 500_000
    assert b.default_round_users(300, 0) == 300 and b.default_round_users(0, 0) == 524288
    assert b.default_round_users(1_250_000, 262144) == 262144
    for n in (1, 7, 524288, 524289, 1_250_000, 9_999_999):
        r = b.default_round_users(n, 0)
        assert r <= 524288 and -(-n // r) == -(-n // 524288)          # as few rounds as 524 288 would take, none longer


def test_usable_cores_is_bounded_by_the_affinity_mask():
    b = _bench()
    n = b.usable_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_gpus_n_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher environment: the script starts the ranks itself (a child
    torch.distributed.run, before it imports torch) and rank 0 prints ONE line with n_gpus = 2.  --dry-run keeps it
    on CPU / gloo with nothing scored: this checks the launch plumbing, not the engine."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--users", "101",
                          "--steps", "3", "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["value"] is None and line["steps"] == 3
    assert line["config"]["shards"] == [[0, 0, 101], [1, 101, 101]]            # rank, user base, users per shard
    # the N > 1 line is self-verifying: two ranks seen, two distinct "devices" (processes here), the backend named
    assert line["ranks_seen"] == 2 and line["distinct_devices"] == 2 and line["dist_backend"] == "gloo"
    assert line["world"]["world_size"] == 2 and len(line["world"]["device_ids"]) == 2
    # a rank that fails makes the parent fail too
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                         env=dict(env, M2D_BENCH_DRYRUN_FAIL_RANK="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300)
    assert bad.returncode != 0


def test_launch_command_is_the_drivers(monkeypatch):
    """The self-launch is the very command the driver uses for N > 1 (task contract), one rank per GPU on 127.0.0.1."""
    b = _bench()
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(b.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5"])
    a = b.parse()
    assert b.launch_ranks(a) == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


# ---- the N > 1 legs of bench.py on CPU: two gloo ranks, an oracle double per shard -------------------------------------
def _legs_worker(rank, world, port, out_dir):
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import random_case
    from test_sharding_gloo import OracleShard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from foodrec_amd.sharding import UserShardedScorer, shard_range
        b = _bench()
        U_per, I, C, E, k = 23, 40, 4, 16, 5
        U = world * U_per
        PM, RE, CE, *_ = random_case(U, I, C, E, 1, seed=3, zero_rows=False)
        base, count = shard_range(U, world, rank)

        class Double(OracleShard):          # what the legs ask of a ScoringEngine beyond the sharding tests' double
            def set_dish_categories(self, cats):
                self.dish_cats = cats.numpy()

            def last_kernel(self):
                return "oracle double"

        dish_cats = np.ones((I, C), np.float32)
        eng = Double(PM, RE, CE, dish_cats, base, count)
        dev = torch.device("cpu")
        # (i) the leg every line carries: n users of each shard, one all-gather
        r1 = b.sharded_topk_leg(torch, dist, eng, U_per, I, C, E, dev, base, 9, world, k=k, repeats=2)
        for key in ("users_per_gpu", "wall_ms_median", "topk_ms_median", "allgather_ms_median", "allgather_bytes_per_rank",
                    "pairs_per_s_whole_job", "own_slice_roundtrip_ok", "kernel"):
            assert key in r1, key
        assert r1["own_slice_roundtrip_ok"] is True and r1["allgather_bytes_per_rank"] == 9 * k * 8
        assert r1["pairs_per_s_whole_job"] > 0
        # (ii) scaling_path's leg: EVERY user of the shard in rounds (here of 7 users: 4 rounds), an all-gather per round issued
        # asynchronously behind the next round's ranking
        sh = UserShardedScorer(eng, U, device=dev, always_collective=True)
        r2 = b.sharded_all_users_leg(torch, dist, sh, I, k, round_users=7, repeats=2)
        for key in ("path", "users_total", "users_per_gpu", "rounds_per_gpu", "wall_ms", "allgather_exposed_ms", "allgather",
                    "allgather_bytes_per_rank", "pairs_per_s_whole_job", "own_slice_roundtrip_ok"):
            assert key in r2, key
        assert r2["path"] == "sharded_topk_allgather" and r2["rounds_per_gpu"] == 4 and r2["users_total"] == U
        assert r2["own_slice_roundtrip_ok"] is True and r2["allgather_bytes_per_rank"] == U_per * k * 8
        # per rank, so that a straggler shows in an N > 1 record
        assert len(r2["shard_ms_per_rank"]) == world and len(r2["allgather_exposed_ms_per_rank"]) == world
        assert r2["allgather_exposed_ms_max"] >= r2["allgather_exposed_ms_min"] >= 0.0
        assert max(r2["shard_ms_per_rank"]) == r2["wall_ms"]
        ident = b.world_identity(torch, dist, dev, world, rank)
        assert ident["ranks_seen"] == world and ident["distinct_devices"] == world and ident["backend"] == "gloo"
        assert ident["world_size"] == world and "rccl_version" in ident
        # what the rounds produce is what one call produces, on every rank
        s_all, i_all = sh.topk_all_users(k, round_users=7)
        s_one, i_one = sh.topk_all_users(k)
        s_two, i_two = sh.topk_all_users(k, round_users=7, pipelined=False)
        assert torch.equal(i_all, i_one) and torch.equal(s_all, s_one) and i_all.shape == (U, k)
        assert torch.equal(i_all, i_two) and torch.equal(s_all, s_two)
        # (iii) pairs routed to the owners of their users
        r3 = b.routed_pairs_leg(torch, dist, eng, U_per, I, C, dev, world, 300, repeats=2)
        for key in ("pairs_per_gpu", "wall_ms_median", "pairs_per_s_whole_job", "own_pairs_match_local_scoring"):
            assert key in r3, key
        assert r3["own_pairs_match_local_scoring"] is True
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_multi_gpu_legs_run_over_gloo_with_an_oracle_double(tmp_path):
    """sharded_topk_leg / sharded_all_users_leg (the `scaling_path` block) / routed_pairs_leg at world 2 on CPU: the keys
    a line carries are there and every rank gets its own slice back through the collectives."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_legs_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(tmp_path, "ok%d" % r)) for r in range(2))


def test_side_leg_watchdog_exits_nonzero():
    """A leg that never returns must not read as a clean run: the line is printed, the exit status is 4 and the record
    names the leg and rank in flight (bench.py::give_up)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os._exit(4)" in src and "os._exit(0)" not in src
    assert '"leg_in_flight": in_flight["leg"]' in src
