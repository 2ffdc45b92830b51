"""User-axis sharding across the GPUs of one node (no reference counterpart -- SURVEY.md section 8e).

Every (user, dish) score depends only on ``Personal_Memory[user]``, ``Recipe_Embedding[dish]``, the
dish's category mask and ``Category_Embedding`` (Model_Recommender.py:57-96), so the path shards
by independent units:

* ``Personal_Memory`` is cut into contiguous user ranges, one per rank (it carries (C+1)/(C+2) of the
  gather bytes and is the only table too large to replicate);
* ``Recipe_Embedding``, ``Category_Embedding`` and the dish masks are replicated;
* one process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI; ``gloo`` in the CPU
  tests).  Pair scoring needs no data-path collective beyond returning ``f32[B]``; retrieval
  all-gathers each shard's *final* per-user top-k (dishes are replicated, so no merge is needed).

The scorer is any object with ``score_pairs(users, items, cats)`` and ``topk_users(users, k)`` working on
tensors of its own device -- in production a ``foodrec_amd.ScoringEngine`` created with
``user_base=base``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(num_users: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous user range ``(base, count)`` of ``rank``: ceil-sized shards, the last may be short."""
    per = -(-num_users // world)
    base = min(rank * per, num_users)
    return base, max(0, min(per, num_users - base))


class UserShardedScorer:
    def __init__(self, scorer, num_users_total: int, group: Optional[dist.ProcessGroup] = None,
                 device: Optional[torch.device] = None):
        self.scorer = scorer
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.num_users_total = int(num_users_total)
        self.base, self.count = shard_range(self.num_users_total, self.world, self.rank)
        self.per = -(-self.num_users_total // self.world)
        self.device = torch.device(device) if device is not None else getattr(scorer, "device", torch.device("cpu"))

    # -- routing ------------------------------------------------------------------------------------
    def owner_of(self, users: torch.Tensor) -> torch.Tensor:
        return torch.div(users.to(torch.int64), self.per, rounding_mode="floor")

    def local_mask(self, users: torch.Tensor) -> torch.Tensor:
        u = users.to(torch.int64)
        return (u >= self.base) & (u < self.base + self.count)

    # -- pair scoring ---------------------------------------------------------------------------------
    def score_pairs(self, users: torch.Tensor, items: torch.Tensor, cats: torch.Tensor) -> torch.Tensor:
        """Every rank passes the SAME batch (global user ids); every rank gets all B scores back.
        Each pair is scored by the one rank that owns its user; the pieces are combined by a sum
        all-reduce over a zero-filled vector (each slot is written by exactly one rank)."""
        B = users.numel()
        if users.numel() and (int(users.min()) < 0 or int(users.max()) >= self.num_users_total):
            raise IndexError("user id out of range [0, %d)" % self.num_users_total)
        mask = self.local_mask(users)
        out = torch.zeros(B, dtype=torch.float32, device=self.device)
        if bool(mask.any()):
            idx = mask.nonzero(as_tuple=True)[0]
            out[idx] = self.scorer.score_pairs(users[idx].contiguous(), items[idx].contiguous(),
                                               cats[idx].contiguous())
        if self.world > 1:
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        return out

    # -- retrieval ------------------------------------------------------------------------------------
    def topk_local(self, k: int, users: Optional[torch.Tensor] = None):
        """Top-k for this shard's users (all of them by default): (scores [n, k], dish ids [n, k])."""
        if users is None:
            users = torch.arange(self.base, self.base + self.count, dtype=torch.int32, device=self.device)
        if users.numel() == 0:
            return (torch.empty((0, k), dtype=torch.float32, device=self.device),
                    torch.empty((0, k), dtype=torch.int32, device=self.device))
        return self.scorer.topk_users(users, k)

    def topk_all_users(self, k: int):
        """Per-user top-k for EVERY user, on every rank: one all-gather of ``[shard, k] x (f32, i32)``.
        Shards are padded to the common size ``per`` for the collective and trimmed afterwards."""
        s, ids = self.topk_local(k)
        if self.world == 1:
            return s, ids
        ps = torch.full((self.per, k), float("nan"), dtype=torch.float32, device=self.device)
        pi = torch.full((self.per, k), -1, dtype=torch.int32, device=self.device)
        ps[: self.count] = s
        pi[: self.count] = ids
        gs = torch.empty((self.world * self.per, k), dtype=torch.float32, device=self.device)
        gi = torch.empty((self.world * self.per, k), dtype=torch.int32, device=self.device)
        dist.all_gather_into_tensor(gs, ps, group=self.group)
        dist.all_gather_into_tensor(gi, pi, group=self.group)
        return gs[: self.num_users_total], gi[: self.num_users_total]
