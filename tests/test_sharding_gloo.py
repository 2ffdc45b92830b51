"""world_size-2 (and 3) CPU tests of the user-sharded path over gloo: routing, the pair-score
all-reduce and the top-k all-gather.  The per-shard scorer is a test double built on the oracle
(the real one is a ScoringEngine on each GPU); what is under test is the host/collective logic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import random_case


class OracleShard:
    """Stands in for ScoringEngine(user_base=base) on one rank."""

    def __init__(self, PM, RE, CE, dish_cats, base, count):
        from oracle import m2d_oracle as oracle
        self.o = oracle
        self.PM, self.RE, self.CE, self.dish_cats = PM[base:base + count], RE, CE, dish_cats
        self.base, self.count = base, count
        self.device = torch.device("cpu")

    def score_pairs(self, users, items, cats):
        u = users.numpy().astype(np.int64) - self.base
        assert u.min() >= 0 and u.max() < self.count, "pair routed to the wrong shard"
        return torch.from_numpy(self.o.inference_f32(self.PM, self.RE, self.CE, u, items.numpy(), cats.numpy()))

    def topk_users(self, users, k):
        u = users.numpy().astype(np.int64) - self.base
        s, i = self.o.topk_catalogue(self.PM, self.RE, self.CE, self.dish_cats, u, k, dtype=np.float32)
        return torch.from_numpy(s.astype(np.float32)), torch.from_numpy(i.astype(np.int32))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, U, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from foodrec_amd.sharding import UserShardedScorer, shard_range
        from oracle import m2d_oracle as oracle
        PM, RE, CE, users, items, cats = random_case(U, 40, 4, 16, 500, seed=3, zero_rows=False)
        dish_cats = np.random.default_rng(4).integers(0, 2, (40, 4)).astype(np.float32)
        dish_cats[dish_cats.sum(1) == 0, 1] = 1
        base, count = shard_range(U, world, rank)
        sh = UserShardedScorer(OracleShard(PM, RE, CE, dish_cats, base, count), U)
        assert (sh.base, sh.count) == (base, count)
        got = sh.score_pairs(torch.from_numpy(users), torch.from_numpy(items), torch.from_numpy(cats))
        ref = oracle.inference_f32(PM, RE, CE, users, items, cats)
        assert np.array_equal(got.numpy(), ref)
        owners = sh.owner_of(torch.from_numpy(users)).numpy()
        assert np.array_equal(owners, users // sh.per)
        s, ids = sh.topk_all_users(5)
        rs, ri = oracle.topk_catalogue(PM, RE, CE, dish_cats, np.arange(U), 5, dtype=np.float32)
        assert s.shape == (U, 5) and np.array_equal(ids.numpy(), ri) and np.array_equal(s.numpy(), rs.astype(np.float32))
        try:
            sh.score_pairs(torch.tensor([U], dtype=torch.int32), torch.tensor([0], dtype=torch.int32), torch.ones(1, 4))
            raise AssertionError("out-of-range user accepted")
        except IndexError:
            pass
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,U", [(2, 37), (3, 10), (2, 1)])
def test_user_sharded_scorer_gloo(tmp_path, world, U):
    mp.spawn(_worker, args=(world, _free_port(), U, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(tmp_path, "ok%d" % r)) for r in range(world))


def test_shard_range_covers_everything():
    from foodrec_amd.sharding import shard_range
    for U in (1, 7, 8, 9, 1000, 10_000_000):
        for w in (1, 2, 4, 8):
            rs = [shard_range(U, w, r) for r in range(w)]
            assert sum(c for _, c in rs) == U
            for (b0, c0), (b1, _) in zip(rs, rs[1:]):
                assert b0 + c0 == b1 or c0 == 0 or b1 == U
