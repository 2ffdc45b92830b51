"""The user-sharded path on the GPU: UserShardedScorer wrapping a REAL ScoringEngine(user_base=...) over the `nccl`
(= RCCL) backend.  A one-GPU box can only form a world of 1, so the scorer is told to issue its collectives anyway
(`always_collective`): all-gather / all-to-all over RCCL with device tensors, the engine's user_base arithmetic, the
collective error check -- the code path the 8-GPU job runs, minus the peers.  World 2 and 3 logic: test_sharding_gloo.py."""
import os
import socket

import numpy as np
import pytest

from helpers import assert_scores_close, random_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world1():
    import torch
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("base", [0, 1000])
def test_sharded_scorer_wraps_a_real_engine(nccl_world1, base):
    """`base` > 0: this rank plays a middle shard -- the engine holds users [base, base + U) of a larger id space, and
    the scorer is given the matching total so that every id routes to this (only) rank."""
    import torch
    from foodrec_amd import ScoringEngine
    from foodrec_amd.sharding import UserShardedScorer
    from oracle import m2d_oracle as oracle
    U, I, C, E, B = 700, 300, 4, 64, 20000
    PM, RE, CE, users, items, cats = random_case(U, I, C, E, B, seed=9 + base)
    dish_cats = np.random.default_rng(1).integers(0, 2, (I, C)).astype(np.float32)
    dish_cats[dish_cats.sum(1) == 0, 2] = 1
    dev = torch.device("cuda", 0)
    eng = ScoringEngine(PM, RE, CE, device=dev, user_base=base)
    eng.set_dish_categories(dish_cats)
    sh = UserShardedScorer(eng, U, device=dev, always_collective=True)
    assert (sh.base, sh.count, sh.per) == (0, U, U)
    # the engine's shard starts at `base`: shift the scorer's view of the id space the same way
    sh.base, sh.num_users_total = base, base + U
    sh.owner_of = lambda u: torch.zeros(u.numel(), dtype=torch.int64, device=u.device)
    t = lambda a: torch.as_tensor(a, device=dev)
    gu = users + base
    ref = oracle.inference_f64(PM, RE, CE, users, items, cats)
    got = sh.score_pairs(t(gu), t(items), t(cats))
    assert_scores_close(got.cpu().numpy(), ref, what="replicated batch")
    got2 = sh.score_pairs_routed(t(gu), t(items), t(cats))
    assert torch.equal(got, got2) or np.array_equal(np.isnan(got.cpu().numpy()), np.isnan(got2.cpu().numpy()))
    assert_scores_close(got2.cpu().numpy(), ref, what="routed batch")
    # retrieval: every user of the shard, all-gathered
    s, ids = sh.topk_all_users(10)
    assert s.shape == (U, 10) and ids.dtype == torch.int32
    rs, ri = oracle.topk_catalogue(PM, RE, CE, dish_cats, np.arange(64), 10)
    # ... in rounds, each round's piece gathered asynchronously (RCCL's stream) behind the next round's kernels: the same bits
    for ru in (U // 3 + 1, 257, U + 5):
        s2, i2 = sh.topk_all_users(10, round_users=ru)
        s3, i3 = sh.topk_all_users(10, round_users=ru, pipelined=False)
        assert torch.equal(i2, ids) and torch.equal(s2, s) and torch.equal(i3, ids) and torch.equal(s3, s), ru
        assert sh.last_allgather_events is not None
    s, ids = s.cpu().numpy(), ids.cpu().numpy()
    for u in range(64):
        full = oracle.inference_f64(PM, RE, CE, np.full(I, u), np.arange(I), dish_cats)
        assert_scores_close(s[u], full[ids[u]], what="top-k scores")
        assert np.sort(full)[-10] <= s[u].min() + 1e-4
    # an id outside the shard is refused through the collective check
    bad = gu.copy(); bad[17] = base + U
    with pytest.raises(IndexError, match="user id %d" % (base + U)):
        sh.score_pairs(t(bad), t(items), t(cats))
    with pytest.raises(IndexError):
        sh.score_pairs_routed(t(bad), t(items), t(cats))
    sh.score_pairs(t(gu[:100]), t(items[:100]), t(cats[:100]))       # and the latch is clear again


def test_routing_is_stable_and_owner_only(nccl_world1):
    """Bucketing on the device: order inside a bucket is the batch order, sizes add up, no host masks."""
    import torch
    from foodrec_amd.sharding import UserShardedScorer
    dev = torch.device("cuda", 0)
    sh = UserShardedScorer(object(), 1000, device=dev)
    sh.world, sh.per = 4, 250                                  # route as a 4-way job would
    users = torch.randint(0, 1000, (5000,), dtype=torch.int32, device=dev)
    order, owner_sorted, counts = sh._bucket(users)
    assert int(counts.sum()) == 5000 and counts.numel() == 4
    own = (users.long() // 250)
    assert torch.equal(owner_sorted, own[order])
    for r in range(4):
        pos = order[owner_sorted == r]
        assert torch.equal(pos, torch.sort(pos).values)        # stable: original order kept
        assert int(counts[r]) == int((own == r).sum())


# ---- two ranks, real engines, ONE GPU, gloo: the N = 2 code path on hardware (correctness only: RCCL wants a GPU per rank) ----
def _two_rank_worker(rank, world, port, out_dir):
    import os
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import foodrec_amd
        from foodrec_amd.sharding import UserShardedScorer, shard_range
        from oracle import m2d_oracle as oracle
        from test_gpu_catalogue import _tables
        U, I, E, k = 5003, 3000, 64, 10                      # 5003 users: the last shard is one user short
        PM, RE, CE, cats = _tables(U, I, 4, E, seed=77, n_nan=3, dup=20)
        base, count = shard_range(U, world, rank)
        dev = torch.device("cuda", 0)
        eng = foodrec_amd.ScoringEngine(PM[base:base + count], RE, CE, device=dev, user_base=base)
        eng.set_dish_categories(cats)
        sh = UserShardedScorer(eng, U, device=dev)
        # retrieval: rounds with the all-gather pipelined behind the next round == rounds + one gather == one call
        s1, i1 = sh.topk_all_users(k, round_users=700)
        s2, i2 = sh.topk_all_users(k, round_users=700, pipelined=False)
        s3, i3 = sh.topk_all_users(k)
        assert i1.shape == (U, k) and torch.equal(i1, i2) and torch.equal(i1, i3) and torch.equal(s1.view(torch.int32), s2.view(torch.int32))
        assert torch.equal(s1.view(torch.int32), s3.view(torch.int32))
        # ... == one engine that holds every user
        full = foodrec_amd.ScoringEngine(PM, RE, CE, device=dev)
        full.set_dish_categories(cats)
        sf, jf = full.topk_users(torch.arange(U, dtype=torch.int32, device=dev), k)
        full.check()
        assert torch.equal(jf, i1) and torch.equal(sf.view(torch.int32), s1.view(torch.int32))
        # pairs: every rank brings its own batch, the owners score, the scores come back (two all-to-alls)
        rng = np.random.default_rng(50 + rank)
        nb = 3000 + 500 * rank
        u2 = rng.integers(0, U, nb).astype(np.int32); d2 = rng.integers(0, I, nb).astype(np.int32)
        m2 = cats[d2]
        got = sh.score_pairs_routed(torch.as_tensor(u2, device=dev), torch.as_tensor(d2, device=dev), torch.as_tensor(m2, device=dev))
        ref = oracle.inference_f64(PM, RE, CE, u2, d2, m2)
        assert_scores_close(got.cpu().numpy(), ref, what="routed pairs over two ranks")
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_over_gloo(tmp_path):
    """N = 2 on hardware, as far as a one-GPU box goes: two processes, each a real ScoringEngine over its user shard on cuda:0,
    torch.distributed over gloo (device tensors).  The sharded retrieval -- rounds, per-round pieces gathered asynchronously --
    returns on every rank the lists of ONE engine holding all users, bit for bit; pairs routed to their owners match the
    restatement.  (RCCL needs a GPU per rank: the N > 1 TIMING stays the driver's.)"""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(tmp_path, "ok%d" % r)) for r in range(2))
