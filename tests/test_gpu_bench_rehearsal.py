"""bench.py's N > 1 line on hardware, as far as a one-GPU box goes: two ranks started the way the driver starts them
(`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 ...`), both on cuda:0 with torch.distributed over gloo
(`M2D_BENCH_REHEARSE_ONE_GPU=1`; RCCL wants a GPU per rank).  The timings mean nothing; what is checked is that every N > 1 leg
runs to the end through the real engine and that the line can be verified from the record alone: `ranks_seen` 2, `distinct_devices`
1 -- which is how a rehearsal, or two ranks that landed on one device by mistake, shows."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_line_over_gloo_on_one_gpu(tmp_path):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = os.path.join(tmp_path, "line.json")
    env = dict(os.environ, M2D_BENCH_REHEARSE_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--users", "100000", "--dishes", "20000", "--pairs", "262144", "--topk-users", "4096", "--scaling-users", "200000",
           "--settle-ms", "20", "--no-cpu-baseline", "--side-timeout", "240", "--out", out]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.load(open(out))
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak" and "rehearsal" in line
    assert line["ranks_seen"] == 2 and line["distinct_devices"] == 1 and line["dist_backend"] == "gloo"
    assert len(line["world"]["device_ids"]) == 2 and line["world"]["device_ids"][0] == line["world"]["device_ids"][1]
    sp = line["scaling_path"]
    assert "error" not in sp and sp["own_slice_roundtrip_ok"] is True and sp["users_total"] == 200000
    assert len(sp["shard_ms_per_rank"]) == 2 and len(sp["allgather_exposed_ms_per_rank"]) == 2
    assert abs(line["topk_path_ms"] / sp["wall_ms"] - 1.0) < 1e-5             # (the tail's copy keeps six significant digits)
    assert line["sharded_topk_allgather"]["own_slice_roundtrip_ok"] is True
    assert line["routed_pairs_alltoall"]["own_pairs_match_local_scoring"] is True
    assert len(line["config"]["workload"]) <= 200
    # the line on stdout is the file's line
    printed = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(printed) == 1 and json.loads(printed[0])["ranks_seen"] == 2


def test_one_rank_bench_line_over_rccl(tmp_path):
    """The launcher path with the REAL backend: one rank under torch.distributed.run, `nccl` (= RCCL) process group bound to its
    device -- every collective the N > 1 legs issue (all-reduce, all-gather, all-gather-object, all-to-all, the asynchronous
    per-round all-gathers) runs through RCCL, at world size 1."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = os.path.join(tmp_path, "line.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "M2D_BENCH_REHEARSE_ONE_GPU"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
           "--users", "100000", "--dishes", "20000", "--pairs", "262144", "--topk-users", "4096", "--scaling-users", "200000",
           "--settle-ms", "20", "--no-cpu-baseline", "--no-config-legs", "--no-projection", "--side-timeout", "240", "--out", out]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.load(open(out))
    assert line["n_gpus"] == 1 and "rehearsal" not in line
    assert line["ranks_seen"] == 1 and line["distinct_devices"] == 1 and line["dist_backend"] == "nccl" and line["rccl_version"]
    assert line["scaling_path"]["own_slice_roundtrip_ok"] is True and line["scaling_path"]["allgather"].startswith("one asynchronous")
    assert line["routed_pairs_alltoall"]["own_pairs_match_local_scoring"] is True
    assert line["sharded_topk_allgather"]["own_slice_roundtrip_ok"] is True
