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
