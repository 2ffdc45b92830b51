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


def test_flags_and_defaults(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.gpus, a.workload, a.users, a.dishes, a.embed, a.pairs) == (1, "pairs", 1_000_000, 100_000, 64, 1 << 22)
    assert a.steps > 0 and a.warmup > 0
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "3"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 7, 3)


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
