"""User-axis sharding across the GPUs of one node (no reference counterpart -- SURVEY.md section 8e).

Every (user, dish) score depends only on ``Personal_Memory[user]``, ``Recipe_Embedding[dish]``, the
dish's category mask and ``Category_Embedding`` (Model_Recommender.py:57-96), so the path shards
by independent units:

* ``Personal_Memory`` is cut into contiguous user ranges, one per rank (it carries (C+1)/(C+2) of the
  gather bytes and is the only table too large to replicate);
* ``Recipe_Embedding``, ``Category_Embedding`` and the dish masks are replicated;
* one process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI; ``gloo`` in the CPU
  tests).

Pair scoring routes every pair to the rank that owns its user (``user // users_per_shard``): pairs are
bucketed by owner with one stable sort, only the owner gathers and scores them, and the only traffic is
ids in / ``f32`` scores out --

* ``score_pairs``        every rank holds the SAME batch (a replicated request): each rank scores its own
                         bucket and one ``all_gather`` of the per-rank score pieces rebuilds ``f32[B]``;
* ``score_pairs_routed`` every rank holds its OWN batch (the serving shape): one ``all_to_all`` carries
                         ``(user, dish, mask)`` records to their owners, one carries the scores back.

Neither masks the full batch on every rank, and the only host round trip per call is the bucket-size vector
(``world`` integers) that sizes the launch and the collective.  Retrieval all-gathers each shard's *final*
per-user top-k (dishes are replicated, so no merge is needed).

The scorer is any object with ``score_pairs(users, items, cats)``, ``topk_users(users, k)`` and ``check()``
working on tensors of its own device -- in production a ``foodrec_amd.ScoringEngine`` created with
``user_base=base`` (``None`` on a rank whose shard is empty).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(num_users: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous user range ``(base, count)`` of ``rank``: ceil-sized shards, the last may be short."""
    per = -(-num_users // world)
    base = min(rank * per, num_users)
    return base, max(0, min(per, num_users - base))


class UserShardedScorer:
    def __init__(self, scorer, num_users_total: int, group: Optional[dist.ProcessGroup] = None,
                 device: Optional[torch.device] = None, always_collective: bool = False):
        """`always_collective`: issue the collectives even at world size 1 (where they are identities) -- lets a
        one-GPU box run the very code path the 8-GPU job runs (tests/test_gpu_sharding.py)."""
        self.scorer = scorer
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.num_users_total = int(num_users_total)
        if self.num_users_total < 1:
            raise ValueError("num_users_total must be positive")
        self.base, self.count = shard_range(self.num_users_total, self.world, self.rank)
        self.per = -(-self.num_users_total // self.world)
        self.device = torch.device(device) if device is not None else getattr(scorer, "device", torch.device("cpu"))
        if self.count > 0 and scorer is None:
            raise ValueError("rank %d owns users [%d, %d) and needs a scorer" % (self.rank, self.base, self.base + self.count))
        self._solo = self.world == 1 and not always_collective

    # -- routing ------------------------------------------------------------------------------------
    def owner_of(self, users: torch.Tensor) -> torch.Tensor:
        """Owning rank of each user id (int64).  An id outside ``[0, num_users_total)`` has no owner: it is sent to
        rank 0, whose engine refuses it (the id error is latched there and raised by ``check()`` on every rank)."""
        u = users.to(torch.int64)
        own = torch.div(u, self.per, rounding_mode="floor")
        return torch.where((u < 0) | (u >= self.num_users_total), torch.zeros_like(own), own)

    def _bucket(self, users: torch.Tensor):
        """Stable bucketing by owner: ``order`` (positions grouped by owner, original order kept inside a bucket),
        ``owner[order]`` and the bucket sizes ``int64[world]`` -- all on the device, no synchronisation."""
        owner = self.owner_of(users)
        # world <= 32767: 16-bit sort keys (a stable radix sort moves a quarter of the key bytes of an int64 sort)
        owner_sorted, order = torch.sort(owner.to(torch.int16), stable=True)
        counts = torch.bincount(owner, minlength=self.world)
        return order, owner_sorted.to(torch.int64), counts

    def _score_local(self, users, items, cats) -> torch.Tensor:
        if users.numel() == 0:
            return torch.empty(0, dtype=torch.float32, device=self.device)
        if self.scorer is None:                     # empty shard: only ownerless ids can land here (rank 0 never is empty)
            return torch.full((users.numel(),), float("nan"), dtype=torch.float32, device=self.device)
        return self.scorer.score_pairs(users.contiguous(), items.contiguous(), cats.contiguous())

    # -- pair scoring ---------------------------------------------------------------------------------
    def score_pairs(self, users: torch.Tensor, items: torch.Tensor, cats: torch.Tensor, check: bool = True) -> torch.Tensor:
        """Every rank passes the SAME batch (global user ids) and gets all B scores back.  Each rank scores only the
        pairs whose user it owns; the per-rank pieces, padded to the largest bucket, travel in one all-gather."""
        B = users.numel()
        C = cats.numel() // max(B, 1)
        out = torch.empty(B, dtype=torch.float32, device=self.device)
        if B == 0:
            return out
        order, owner_sorted, counts = self._bucket(users)
        counts_h: List[int] = counts.tolist()                         # the one host round trip: sizes the launch
        off_h = [0]
        for c in counts_h:
            off_h.append(off_h[-1] + c)
        lo, n = off_h[self.rank], counts_h[self.rank]
        idx = order[lo:lo + n]
        local = self._score_local(users[idx], items[idx], cats.reshape(B, C)[idx])
        if self._solo:
            out[idx] = local
        else:
            cmax = max(counts_h)
            piece = torch.zeros(cmax, dtype=torch.float32, device=self.device)
            piece[:n] = local
            gathered = torch.empty(self.world * cmax, dtype=torch.float32, device=self.device)
            dist.all_gather_into_tensor(gathered, piece, group=self.group)
            # sorted position p of bucket r sits at gathered[r * cmax + (p - off[r])]
            off = torch.cumsum(counts, 0) - counts
            sel = owner_sorted * cmax + (torch.arange(B, device=self.device) - off[owner_sorted])
            out[order] = gathered[sel]
        if check:
            self.check()
        return out

    def score_pairs_routed(self, users: torch.Tensor, items: torch.Tensor, cats: torch.Tensor,
                           check: bool = True) -> torch.Tensor:
        """Every rank passes its OWN batch (global user ids, any owners) and gets its own scores back: one all-to-all
        of ``[user, dish, mask bits]`` int32 records to the owners, the owners score, one all-to-all of ``f32`` back."""
        B = users.numel()
        C = cats.shape[-1] if cats.dim() > 1 else (cats.numel() // max(B, 1))
        if self._solo:
            out = self._score_local(users, items, cats.reshape(B, C))
            if check:
                self.check()
            return out
        order, _, send_counts = self._bucket(users)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        sc: List[int] = send_counts.tolist()                          # the one host round trip (two small vectors)
        rc: List[int] = recv_counts.tolist()
        rec = torch.empty((B, 2 + C), dtype=torch.int32, device=self.device)        # one record per pair, gathered once
        rec[:, 0] = users
        rec[:, 1] = items
        rec[:, 2:] = cats.reshape(B, C).to(torch.float32).view(torch.int32)
        rec = rec[order]
        got = torch.empty((sum(rc), 2 + C), dtype=torch.int32, device=self.device)
        dist.all_to_all_single(got, rec, rc, sc, group=self.group)
        scores = self._score_local(got[:, 0].contiguous(), got[:, 1].contiguous(),
                                   got[:, 2:].contiguous().view(torch.float32))
        back = torch.empty(B, dtype=torch.float32, device=self.device)
        dist.all_to_all_single(back, scores, sc, rc, group=self.group)
        out = torch.empty(B, dtype=torch.float32, device=self.device)
        out[order] = back
        if check:
            self.check()
        return out

    def check(self):
        """Collective: synchronise, and raise ``IndexError`` on EVERY rank when any rank's engine latched an
        out-of-range id (TF-CPU's GatherV2 would have raised for the whole call)."""
        msg = ""
        if self.scorer is not None and hasattr(self.scorer, "check"):
            try:
                self.scorer.check()
            except IndexError as e:
                msg = str(e) or "id out of range"
        if not self._solo:
            flag = torch.tensor([1 if msg else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            if int(flag.item()) and not msg:
                msg = "an id was out of range on another rank"
        if msg:
            raise IndexError(msg)

    # -- retrieval ------------------------------------------------------------------------------------
    def topk_local(self, k: int, users: Optional[torch.Tensor] = None):
        """Top-k for this shard's users (all of them by default): (scores [n, k], dish ids [n, k])."""
        if users is None:
            users = torch.arange(self.base, self.base + self.count, dtype=torch.int32, device=self.device)
        if users.numel() == 0:
            return (torch.empty((0, k), dtype=torch.float32, device=self.device),
                    torch.empty((0, k), dtype=torch.int32, device=self.device))
        return self.scorer.topk_users(users, k)

    def topk_local_rounds(self, k: int, round_users: int = 524288):
        """Top-k for EVERY user of this shard, computed in rounds of `round_users` users -- a launch and its scratch stay
        the same size whatever the shard holds (1.25 M users per GPU at BASELINE configs[3]) -- into one pair of
        ``[count, k]`` buffers: (scores f32, dish ids i32).  A round is one retrieval call, and a call sorts ITS users by
        the mask patterns that can reach their top-k: larger rounds make blocks of users that share more (fewer tiles
        stepped through) and spread the per-call launches thinner -- 10 M users x 1 M dishes: 0.59 s in rounds of 65 536,
        0.46 s in rounds of 262 144, 0.44 s at 524 288 (round 4, with the left-out records: 0.418 / 0.406 s; the scratch grows
        with the round: about 1 KB per user, 0.5 GB at the default)."""
        s = torch.empty((self.count, k), dtype=torch.float32, device=self.device)
        ids = torch.empty((self.count, k), dtype=torch.int32, device=self.device)
        into = getattr(self.scorer, "topk_users_into", None)          # ScoringEngine: straight into the slices, no copy
        for lo in range(0, self.count, int(round_users)):
            n = min(int(round_users), self.count - lo)
            users = torch.arange(self.base + lo, self.base + lo + n, dtype=torch.int32, device=self.device)
            if into is not None:
                into(users, k, s[lo:lo + n], ids[lo:lo + n])
            else:
                rs, ri = self.scorer.topk_users(users, k)
                s[lo:lo + n] = rs
                ids[lo:lo + n] = ri
        return s, ids

    def _gather_topk(self, s: torch.Tensor, ids: torch.Tensor, rows: int, k: int):
        """One all-gather of ``[rows, k] x (f32 score, i32 id)`` per rank; scores and ids travel in one int32 buffer.
        Ranks with fewer than `rows` results pad with (NaN, -1)."""
        n = s.shape[0]
        piece = torch.empty((2, rows, k), dtype=torch.int32, device=self.device)
        if n < rows:
            piece[0].view(torch.float32).fill_(float("nan"))
            piece[1].fill_(-1)
        piece[0, :n] = s.view(torch.int32)
        piece[1, :n] = ids
        gathered = torch.empty((self.world * 2, rows, k), dtype=torch.int32, device=self.device)
        dist.all_gather_into_tensor(gathered, piece, group=self.group)
        gathered = gathered.view(self.world, 2, rows, k)
        return gathered[:, 0].reshape(self.world * rows, k).view(torch.float32), gathered[:, 1].reshape(self.world * rows, k)

    def topk_users_gathered(self, users: torch.Tensor, k: int):
        """Top-k for `users` (ids of THIS shard; the same count on every rank), all-gathered: every rank receives
        ``[world * n, k]`` scores and dish ids, rank r's block at rows ``[r * n, (r + 1) * n)``."""
        s, ids = self.topk_local(k, users)
        if self._solo:
            return s, ids
        return self._gather_topk(s, ids, users.numel(), k)

    def topk_all_users(self, k: int, round_users: Optional[int] = None, pipelined: Optional[bool] = None):
        """Per-user top-k for EVERY user, on every rank: ``[num_users_total, k]`` scores and dish ids, rank r's users at rows
        ``[r * per, r * per + count_r)``.  Shards are padded to the common size ``per`` for the collective and trimmed
        afterwards.  `round_users`: the local part runs in rounds of that many users.  With rounds the all-gather is
        pipelined (`pipelined`, default on): round r's ``[rows, k] x (f32, i32)`` piece is gathered asynchronously while
        round r + 1 is being ranked, so only the last round's exchange is exposed; `pipelined=False` ranks the whole shard
        first and gathers once (the same result, bit for bit -- tests/test_sharding_gloo.py)."""
        if pipelined is None:
            pipelined = bool(round_users)
        if self._solo or not (pipelined and round_users):
            s, ids = self.topk_local_rounds(k, round_users) if round_users else self.topk_local(k)
            if self._solo:
                return s, ids
            gs, gi = self._gather_topk(s, ids, self.per, k)
            return gs[: self.num_users_total], gi[: self.num_users_total]
        return self._topk_all_users_pipelined(k, int(round_users))

    def _topk_all_users_pipelined(self, k: int, round_users: int):
        """Rounds of `round_users` rows of the padded shard (every rank runs the same number of rounds of the same sizes, an
        empty or short shard pads with (NaN, -1)): rank the round into this rank's piece, hand the piece to an asynchronous
        all-gather -- the collective's stream waits for the ranking kernels queued so far and runs beside the next round's --
        and, one round later, copy the gathered round into the result.  Two staging buffers alternate; the copy of round
        r - 1 is queued (behind its all-gather, in front of round r + 1's kernels) before round r + 1 may overwrite a piece."""
        world, per, dev = self.world, self.per, self.device
        if per == 0:                                            # no user anywhere: nothing to rank, nothing to exchange
            self.last_allgather_events = None
            return (torch.empty((0, k), dtype=torch.float32, device=dev), torch.empty((0, k), dtype=torch.int32, device=dev))
        out_s = torch.empty((world, per, k), dtype=torch.float32, device=dev)
        out_i = torch.empty((world, per, k), dtype=torch.int32, device=dev)
        rows_max = min(round_users, per)
        pieces = [torch.empty((2, rows_max, k), dtype=torch.int32, device=dev) for _ in range(2)]
        stages = [torch.empty((world * 2, rows_max, k), dtype=torch.int32, device=dev) for _ in range(2)]     # rank-major pieces
        into = getattr(self.scorer, "topk_users_into", None) if self.scorer is not None else None
        pending = None                                          # (work, stage, lo, rows) of the round in flight
        self.last_allgather_events = None

        def finish(p):
            work, stage, lo, rows = p
            work.wait()                                         # the current stream waits for the collective; the host does not
            stage = stage.view(world, 2, rows, k)
            out_s[:, lo:lo + rows] = stage[:, 0].view(torch.float32)
            out_i[:, lo:lo + rows] = stage[:, 1]

        for r, lo in enumerate(range(0, per, round_users)):
            rows = min(round_users, per - lo)                   # the same on every rank
            n = max(0, min(rows, self.count - lo))              # rows of this round that hold users of this shard
            piece, stage = pieces[r & 1], stages[r & 1]
            ps, pi = piece[0, :rows].view(torch.float32), piece[1, :rows]
            if n < rows:
                ps[n:] = float("nan")
                pi[n:] = -1
            if n > 0:
                users = torch.arange(self.base + lo, self.base + lo + n, dtype=torch.int32, device=dev)
                if into is not None:
                    into(users, k, ps[:n], pi[:n])
                else:
                    rs, ri = self.scorer.topk_users(users, k)
                    ps[:n] = rs
                    pi[:n] = ri
            if pending is not None:
                finish(pending)
            src = piece if rows == rows_max else piece[:, :rows].contiguous()
            dst = stage if rows == rows_max else torch.empty((world * 2, rows, k), dtype=torch.int32, device=dev)
            work = dist.all_gather_into_tensor(dst, src, group=self.group, async_op=True)
            pending = (work, dst, lo, rows)
        if dev.type == "cuda":                                  # bench.py: what of the exchange is NOT hidden behind ranking
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            finish(pending)
            ev1.record()
            self.last_allgather_events = (ev0, ev1)
        else:
            finish(pending)
        return (out_s.view(world * per, k)[: self.num_users_total], out_i.view(world * per, k)[: self.num_users_total])
