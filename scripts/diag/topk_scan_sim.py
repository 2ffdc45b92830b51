"""Simulation of the pattern-grouped retrieval scan's candidate tiles (numpy, CPU): for one wave (32 users x 2 lanes,
lane h of a user owns rows 4h + (r & 3) + 8 (r >> 2) of every 32-row tile) count the tiles in which some lane holds a
score >= its threshold, for several scan orders / threshold rules.  N(0, 1/E) tables, uniform non-empty masks.
    python scripts/diag/topk_scan_sim.py [dishes] [E]"""
import sys
import numpy as np

I = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 64
C, k, NU = 4, 10, 32
rng = np.random.default_rng(0)
s = E ** -0.5
RE = (rng.standard_normal((I, E)) * s).astype(np.float32)
PM = (rng.standard_normal((NU, C + 1, E)) * s).astype(np.float32)
CE = (rng.standard_normal((C, E)) * s).astype(np.float32)
pat = rng.integers(1, 16, I)
cats = ((pat[:, None] >> np.arange(C)) & 1).astype(np.float32)
n = cats.sum(1)
a, b = np.float32(0.99), np.float32(1) - np.float32(0.99)
hc = PM[:, 0] @ CE.T                                     # [NU, C]
alpha = a * (hc @ cats.T) / n                            # [NU, I]
low = np.einsum("uce,ie,ic->ui", PM[:, 1:], RE, cats) / n
score = alpha + b * low                                  # [NU, I]
norm = np.linalg.norm(RE, axis=1)
mean, sd = norm.mean(), norm.std()
bucket = np.clip(((mean + 2 * sd - norm) * (16 / (4 * sd))).astype(int), 0, 15)


def order_tiles(keys):
    """dishes sorted by key tuple; every distinct leading group padded to whole tiles of 32 (-1 = padding)"""
    order = np.lexsort(keys[::-1])
    grp = keys[0][order] if len(keys) == 1 else np.stack([kk[order] for kk in keys[:-2]] + [keys[-2][order] * 0], 0).T
    return order


def tiles_for(group_key, inner_keys):
    out = []
    for g in np.unique(group_key):
        idx = np.flatnonzero(group_key == g)
        idx = idx[np.lexsort([kk[idx] for kk in inner_keys][::-1])]
        pad = (-len(idx)) % 32
        out.append(np.concatenate([idx, -np.ones(pad, dtype=np.int64)]))
    return np.concatenate(out).reshape(-1, 32)


rows_of = [np.array([4 * h + (r & 3) + 8 * (r >> 2) for r in range(16)]) for h in (0, 1)]


# rigorous seed: every dish of pattern P scores >= alpha_P - |w_P| max|r|; with >= k dishes in P that bounds the final k-th
wP = np.zeros((NU, 16, E), np.float32)
aP = np.full((NU, 16), -np.inf, np.float32)
for q in range(1, 16):
    cs = [c for c in range(C) if (q >> c) & 1]
    wP[:, q] = b * PM[:, 1:][:, cs].sum(1) / len(cs)
    aP[:, q] = a * hc[:, cs].sum(1) / len(cs)
rmax = np.array([norm[pat == q].max() if (pat == q).sum() >= k else np.inf for q in range(16)])
seed = np.where(np.isfinite(rmax[None, 1:]), aP[:, 1:] - np.linalg.norm(wP[:, 1:], axis=2) * np.where(np.isfinite(rmax[1:]), rmax[1:], 0)[None, :], -np.inf).max(1).astype(np.float32)


def simulate(tiles, rule):
    cand_tiles = 0
    lists = np.full((NU, 2, k), -np.inf, dtype=np.float32)
    thr = np.full((NU, 2), -np.inf, dtype=np.float32)
    if rule.endswith("+seed"):
        thr = np.repeat(seed[:, None], 2, 1).copy()
        rule = rule[:-5]
    thr0 = thr.copy()
    ins = 0
    for t in tiles:
        valid = t >= 0
        sc = np.where(valid[None, :], score[:, np.maximum(t, 0)], -np.inf)   # [NU, 32]
        any_c = False
        for h in (0, 1):
            v = sc[:, rows_of[h]]                         # [NU, 16]
            c = v >= thr[:, h][:, None]
            c &= np.isfinite(v)
            if c.any():
                any_c = True
                for u in np.flatnonzero(c.any(1)):
                    merged = np.sort(np.concatenate([lists[u, h], v[u][c[u]]]))[::-1][:k]
                    ins += int(c[u].sum())
                    lists[u, h] = merged
        if any_c:
            cand_tiles += 1
            own = lists[:, :, k - 1]
            if rule == "own":
                thr = own.copy()
            elif rule == "max":
                thr = np.repeat(own.max(1)[:, None], 2, 1)
            elif rule == "mid":
                mid = np.minimum(lists[:, 0, k // 2 - 1], lists[:, 1, k // 2 - 1])
                thr = np.repeat(np.maximum(own.max(1), mid)[:, None], 2, 1)
            elif rule == "exact":
                m = np.sort(lists.reshape(NU, -1), axis=1)[:, ::-1][:, k - 1]
                thr = np.repeat(m[:, None], 2, 1)
            thr = np.maximum(thr, thr0)
    return cand_tiles / len(tiles), ins / (NU * 2)


dish = np.arange(I)
orders = {
    "pattern, bucket, id (now)": tiles_for(pat, [bucket, dish]),
    "phase(bucket<3), pattern, bucket, id": tiles_for((bucket >= 3) * 16 + pat, [bucket, dish]),
}
for name, tl in orders.items():
    for rule in ("max", "mid", "max+seed", "mid+seed"):
        f, ins = simulate(tl, rule)
        print("%-40s thr=%-5s tiles %5d  candidate tiles %.3f  insertions per lane %.1f" % (name, rule, len(tl), f, ins))

# One pattern only (what a block of a pruned launch steps through: its users' best pattern), 32 users whose best pattern it
# is, by threshold rule -- "exact" is the k-th largest of the two lanes' lists together, max_i min(a_i, b_{k-i}).
if len(sys.argv) > 3:
    q = int(sys.argv[3])
    pool = (rng.standard_normal((4000, C + 1, E)) * s).astype(np.float32)
    hcp = pool[:, 0] @ CE.T
    aPp = np.full((4000, 16), -np.inf, np.float32)
    for qq in range(1, 16):
        cs = [c for c in range(C) if (qq >> c) & 1]
        aPp[:, qq] = a * hcp[:, cs].sum(1) / len(cs)
    pick = np.flatnonzero(aPp[:, 1:].argmax(1) + 1 == q)[:NU]
    PM = pool[pick]
    hc = PM[:, 0] @ CE.T
    alpha = a * (hc @ cats.T) / n
    low = np.einsum("uce,ie,ic->ui", PM[:, 1:], RE, cats) / n
    score = alpha + b * low
    cs = [c for c in range(C) if (q >> c) & 1]
    wq = b * PM[:, 1:][:, cs].sum(1) / len(cs)
    aq = a * hc[:, cs].sum(1) / len(cs)
    sub = dish[pat == q]
    tl = tiles_for(pat[pat == q] * 0, [bucket[pat == q], dish[pat == q]])
    tl = np.where(tl >= 0, sub[np.maximum(tl, 0)], -1)
    # scan-start bound: the k-th largest exact score among the pattern's first 16 rows (the plan's probe rows)
    first = tl.reshape(-1)[:16]
    seed = np.sort(score[:, first], axis=1)[:, -k].astype(np.float32)
    print("pattern %d (%d dishes, %d tiles), %d users whose best pattern it is" % (q, len(sub), len(tl), len(pick)))
    for rule in ("own+seed", "max+seed", "mid+seed", "exact+seed"):
        f, ins = simulate(tl, rule)
        print("  thr=%-10s candidate tiles %.3f  insertions per lane %.1f" % (rule, f, ins))
