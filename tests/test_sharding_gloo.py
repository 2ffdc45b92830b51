"""world_size-2 (and 3) CPU tests of the user-sharded path over gloo: owner bucketing, the replicated-batch
all-gather, the routed all-to-all and the top-k all-gather.  The per-shard scorer is a test double built on the
oracle (the real one is a ScoringEngine on each GPU -- tests/test_gpu_sharding.py runs that); what is under test
here is the host / collective logic."""
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

        self.pending = None             # like the engine: a bad id is latched, NaN is written, check() raises
        self.pairs_scored = 0

    def score_pairs(self, users, items, cats):
        u = users.numpy().astype(np.int64) - self.base
        self.pairs_scored += len(u)
        bad = (u < 0) | (u >= self.count)
        if bad.any() and self.pending is None:
            self.pending = "user id %d is out of range" % int(users.numpy()[np.flatnonzero(bad)[0]])
        out = np.full(len(u), np.nan, np.float32)
        ok = ~bad
        out[ok] = self.o.inference_f32(self.PM, self.RE, self.CE, u[ok], items.numpy()[ok], cats.numpy()[ok])
        return torch.from_numpy(out)

    def check(self):
        msg, self.pending = self.pending, None
        if msg:
            raise IndexError(msg)

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
        shard = OracleShard(PM, RE, CE, dish_cats, base, count) if count else None     # an empty shard has no engine
        sh = UserShardedScorer(shard, U)
        assert (sh.base, sh.count) == (base, count)
        got = sh.score_pairs(torch.from_numpy(users), torch.from_numpy(items), torch.from_numpy(cats))
        ref = oracle.inference_f32(PM, RE, CE, users, items, cats)
        assert np.array_equal(got.numpy(), ref)
        owners = sh.owner_of(torch.from_numpy(users)).numpy()
        assert np.array_equal(owners, users // sh.per)
        # only the owner scored a pair: this rank saw exactly its bucket, not the whole batch
        mine = int((owners == rank).sum())
        assert (shard.pairs_scored if shard else 0) == mine
        # routed form: every rank brings its OWN batch (different sizes, any owners) and gets its own scores back
        rng = np.random.default_rng(100 + rank)
        nb = 50 + 37 * rank
        u2 = rng.integers(0, U, nb).astype(np.int32)
        d2 = rng.integers(0, 40, nb).astype(np.int32)
        m2 = rng.integers(0, 2, (nb, 4)).astype(np.float32); m2[m2.sum(1) == 0, 0] = 1
        got2 = sh.score_pairs_routed(torch.from_numpy(u2), torch.from_numpy(d2), torch.from_numpy(m2))
        assert np.array_equal(got2.numpy(), oracle.inference_f32(PM, RE, CE, u2, d2, m2))
        empty = sh.score_pairs_routed(torch.zeros(0, dtype=torch.int32), torch.zeros(0, dtype=torch.int32), torch.zeros((0, 4)))
        assert empty.numel() == 0
        s, ids = sh.topk_all_users(5)
        rs, ri = oracle.topk_catalogue(PM, RE, CE, dish_cats, np.arange(U), 5, dtype=np.float32)
        assert s.shape == (U, 5) and np.array_equal(ids.numpy(), ri) and np.array_equal(s.numpy(), rs.astype(np.float32))
        # rounds with the all-gather pipelined behind the next round's ranking (an uneven last round, rounds larger than the
        # shard, a last shard that is short or empty) == rounds with one gather at the end == no rounds at all
        for ru in (1, 2, 3, 4, 7, 64):
            s2, i2 = sh.topk_all_users(5, round_users=ru)
            s3, i3 = sh.topk_all_users(5, round_users=ru, pipelined=False)
            for a, b in ((s2, s), (s3, s)):
                assert a.shape == (U, 5) and np.array_equal(a.numpy(), b.numpy(), equal_nan=True)
            assert np.array_equal(i2.numpy(), ids.numpy()) and np.array_equal(i3.numpy(), ids.numpy())
        # an id no shard owns: refused on EVERY rank (collective check), in both forms
        for bad_user in (U, -1, 10 * U + 7):
            for fn in (sh.score_pairs, sh.score_pairs_routed):
                try:
                    fn(torch.tensor([0, bad_user], dtype=torch.int32), torch.tensor([0, 0], dtype=torch.int32), torch.ones(2, 4))
                    raise AssertionError("out-of-range user accepted")
                except IndexError:
                    pass
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,U", [(2, 37), (3, 10), (2, 1), (3, 13)])
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
