"""bench.py's flags, the self-launch of N ranks and the CPU dry run of the launch plumbing."""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys

WORKLOAD_HELP = (
    "pairs = BASELINE configs[1] (reference forward, HBM-bound); ingredients = configs[1] with the build-defined "
    "10k-row ingredient table on the high-level path; mlp = configs[2] (build-defined 3-layer head, MFMA-bound; pass "
    "--embed 128); topk = configs[3]/[4] retrieval: full-catalogue top-10 for --topk-users users per GPU + all-gather "
    "of the results (MFMA-bound); train = the reference's training step (SURVEY.md 8f N4) at its own default sizes "
    "unless --users/--dishes/--embed/--pairs are given, single GPU")


def parse(argv=None):
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
                   help=WORKLOAD_HELP)
    p.add_argument("--ingredients", type=int, default=10_000, help="rows of the ingredient table")
    p.add_argument("--config", type=int, choices=[3, 4], default=None,
                   help="BASELINE.json configs[3] / configs[4] as the timed step: 10 M / N users per GPU x 1 M "
                        "replicated dishes, E = 64 / 128, top-10 for EVERY user of the shard in rounds of "
                        "--round-users, then ONE all-gather of [shard, 10] x (f32 score, i32 id)")
    p.add_argument("--round-users", type=int, default=0,
                   help="users per retrieval launch in the sharded top-k path (0 = the shard in the fewest even "
                        "rounds of at most 524288)")
    p.add_argument("--no-projection", action="store_true",
                   help="skip scaling_path.projected_world8 (the N = 8 per-GPU shape timed on one GPU)")
    p.add_argument("--scaling-users", type=int, default=10_000_000,
                   help="users over ALL GPUs in the scaling_path block (configs[3]: 10 M; 0 = leave the block out)")
    p.add_argument("--topk-weighted-masks", action="store_true",
                   help="--workload topk with category weights other than 0 / 1 (the placeholder is float, "
                        "Model_Recommender.py:32): the dense exact-f32 kernel m2d_topk_mfma serves the call")
    p.add_argument("--topk-k", type=int, default=10, help="--workload topk: list length")
    p.add_argument("--topk-with-ingredients", action="store_true",
                   help="workload topk: set the ingredient table first (retrieval over [H[d] | RE[d]] rows)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg")
    p.add_argument("--topk-users", type=int, default=65536, help="users in the catalogue top-k side leg (0 = skip)")
    p.add_argument("--no-side", action="store_true", help="skip every leg outside the timed region")
    p.add_argument("--no-config-legs", action="store_true",
                   help="skip the BASELINE configs[2] / configs[4] side legs of the default run (cfg2_* / cfg4_*)")
    p.add_argument("--unique-users", action="store_true",
                   help="profiling aid: every user at most once per step (pairs <= users), no table reuse")
    p.add_argument("--sweep", action="store_true", help="also time the kernel knobs (stderr only)")
    p.add_argument("--opt", action="append", default=[], help="engine option name=value")
    p.add_argument("--out", default=None, help="also write the JSON line to this file (rank 0)")
    p.add_argument("--settle-ms", type=float, default=150.0,
                   help="before the W warmup steps: run the step untimed for this long so that the part's clocks have "
                        "ramped (the first ~25 launches after an idle period run 5-7 %% slow); not counted as steps, "
                        "reported in the line as config.settle; 0 = off")
    p.add_argument("--side-timeout", type=float, default=420.0,
                   help="seconds the legs after the timed region may take before rank 0 prints the headline line "
                        "without them and every rank exits")
    p.add_argument("--dry-run", action="store_true",
                   help="launch plumbing only (CPU, gloo): the ranks rendezvous, exchange their shard ranges and "
                        "rank 0 prints a line with value null -- nothing is scored, no GPU is touched")
    return p.parse_args(argv)


def launch_ranks(a, script):
    """`python bench.py --gpus N` with N > 1: start one rank per GPU as fresh child processes through
    torch.distributed.run and relay their status.  Runs before torch is imported: this process never touches a
    GPU (and never exec-replaces itself)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % a.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def emit(line, out_path=None):
    """Rank 0: the ONE JSON line on stdout, and the same text in --out (a file is not preceded by RCCL's banner)."""
    text = json.dumps(line)
    print(text)
    sys.stdout.flush()
    if out_path:
        with open(out_path, "w") as f:
            f.write(text + "\n")


def dry_run(a):
    """--dry-run: the multi-process plumbing of bench.py on CPU -- rendezvous, shard ranges, one collective, the
    world identity block, the rank-0 line -- with no engine and no scoring."""
    import torch
    import torch.distributed as dist
    from foodrec_amd.sharding import shard_range
    from .sharded import world_identity
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if os.environ.get("M2D_BENCH_DRYRUN_FAIL_RANK") == str(rank):     # test seam: a dying rank must fail the launcher
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
    ident = world_identity(torch, dist if world > 1 else None, torch.device("cpu"), world, rank)
    if rank == 0:
        emit({"metric": "scored (user,dish) pairs/sec", "value": None, "unit": "pairs/s", "n_gpus": world,
              "steps": a.steps, "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none (dry run)", "dry_run": True,
              "config": {"workload": "launch plumbing only", "shards": allr.view(world, 3).tolist()},
              "world": ident, **ident_scalars(ident)}, a.out)
    if world > 1:
        dist.destroy_process_group()


def ident_scalars(ident):
    """The world identity as top-level scalars (a record that keeps only scalar keys keeps these)."""
    return {"ranks_seen": ident["ranks_seen"], "distinct_devices": ident["distinct_devices"],
            "dist_backend": ident["backend"], "rccl_version": ident["rccl_version"]}
