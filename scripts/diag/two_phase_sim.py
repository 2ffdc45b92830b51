"""Would a second look at the plan pay?  Phase 1: every user's best pattern only (the pattern whose lower bound is the scan-start
bound); its exact k-th score is then a far better threshold than the bound, and phase 2 needs only the patterns whose UPPER bound
still reaches it.  Tiles a block steps through (users sorted by mask in each phase, blocks of 256), against the one-phase plan.
numpy, CPU.   python scripts/diag/two_phase_sim.py [users] [dishes] [E]"""
import sys
import numpy as np

NU = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
I = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
E = int(sys.argv[3]) if len(sys.argv) > 3 else 64
C, k = 4, 10
rng = np.random.default_rng(0)
s = E ** -0.5
RE = (rng.standard_normal((I, E)) * s).astype(np.float32)
PM = (rng.standard_normal((NU, C + 1, E)) * s).astype(np.float32)
CE = (rng.standard_normal((C, E)) * s).astype(np.float32)
pat = rng.integers(1, 16, I)
a, b = np.float32(0.99), np.float32(1) - np.float32(0.99)
norm = np.linalg.norm(RE, axis=1)
rows = np.array([(pat == q).sum() for q in range(16)])
rmax = np.array([norm[pat == q].max() if rows[q] else 0.0 for q in range(16)])
hc = PM[:, 0] @ CE.T
aP = np.full((NU, 16), -np.inf, np.float32)
reach = np.zeros((NU, 16), np.float32)
W = np.zeros((16, NU, E), np.float32)
for q in range(1, 16):
    cs = [c for c in range(C) if (q >> c) & 1]
    aP[:, q] = a * hc[:, cs].sum(1) / len(cs)
    W[q] = b * PM[:, 1:][:, cs].sum(1) / len(cs)
    reach[:, q] = np.linalg.norm(W[q], axis=1) * rmax[q]
lo = np.where(rows[None, :] >= k, aP - reach * 1.0001, -np.inf)
hi = np.where(rows[None, :] > 0, aP + reach * 1.0001, -np.inf)
seed = lo[:, 1:].max(1)
p1 = lo[:, 1:].argmax(1) + 1
rel = hi >= seed[:, None]
rel[:, 0] = False
tiles = -(-rows // 32)

def scanned(masks_bool):
    key = (masks_bool * (1 << np.arange(16))[None, :]).sum(1)
    order = np.argsort(key, kind="stable")
    tot = 0
    for b0 in range(0, NU, 256):
        u = order[b0:b0 + 256]
        u = u[key[u] > 0]                                   # users with nothing left to scan take no block
        if len(u):
            tot += tiles[masks_bool[u].any(0)].sum()
    nblk = -(-int((key > 0).sum()) // 256)
    return tot, nblk

one, nb1 = scanned(rel)
full = (NU // 256) * tiles.sum()
print("one phase: %.3f of all tiles" % (one / full))
# phase 1: the best pattern alone; its exact k-th score
m1 = np.zeros((NU, 16), bool); m1[np.arange(NU), p1] = True
t1, _ = scanned(m1)
thr1 = np.empty(NU, np.float32)
for q in range(1, 16):
    us = np.flatnonzero(p1 == q)
    if len(us) == 0: continue
    sc = aP[us, q][:, None] + W[q][us] @ RE[pat == q].T
    thr1[us] = np.sort(sc, axis=1)[:, -k]
m2 = (hi >= thr1[:, None] * (1 - 1e-6) - 1e-6) & ~m1
m2[:, 0] = False
t2, nb2 = scanned(m2)
print("two phases: %.3f + %.3f = %.3f of all tiles; users that need phase 2: %.3f (%d blocks); patterns left per such user %.2f" %
      (t1 / full, t2 / full, (t1 + t2) / full, m2.any(1).mean(), nb2, m2.sum(1)[m2.any(1)].mean() if m2.any() else 0))

# A cheaper middle: a better scan-start bound from a few PROBE dishes of the best pattern (the k-th largest exact score among the
# pattern's n highest-norm dishes is a valid lower bound of the pattern's k-th score), one phase as now.
for nprobe in (16, 32, 64, 128, 256):
    thr0 = seed.copy()
    for q in range(1, 16):
        us = np.flatnonzero(p1 == q)
        if len(us) == 0: continue
        d = np.flatnonzero(pat == q)
        d = d[np.argsort(-norm[d])[:nprobe]]
        sc = aP[us, q][:, None] + W[q][us] @ RE[d].T
        thr0[us] = np.maximum(seed[us], np.sort(sc, axis=1)[:, -k] * (1 - 1e-6) - 1e-6)
    relp = hi >= thr0[:, None]
    relp[:, 0] = False
    t, _ = scanned(relp)
    print("probe %3d dishes: %.3f of all tiles, %.2f patterns per user" % (nprobe, t / full, relp.sum(1).mean()))

# Sort keys for the one-phase plan with 16 probe rows: the mask's value (now) against (best pattern, mask)
thr0 = seed.copy()
for q in range(1, 16):
    us = np.flatnonzero(p1 == q)
    if len(us) == 0: continue
    d = np.flatnonzero(pat == q)
    d = d[np.argsort(-norm[d])[:16]]
    sc = aP[us, q][:, None] + W[q][us] @ RE[d].T
    thr0[us] = np.maximum(seed[us], np.sort(sc, axis=1)[:, -k] * (1 - 1e-6) - 1e-6)
relp = hi >= thr0[:, None]
relp[:, 0] = False
maskv = (relp * (1 << np.arange(16))[None, :]).sum(1)
def scanned_key(key):
    order = np.argsort(key, kind="stable")
    return sum(tiles[relp[order[b0:b0 + 256]].any(0)].sum() for b0 in range(0, NU, 256))
npat = relp.sum(1)
for name, key in (("mask value", maskv), ("(best pattern, mask)", p1 * 65536 + maskv), ("(patterns, mask)", npat * 65536 + maskv),
                  ("(best pattern, patterns, mask)", (p1 * 16 + npat) * 65536 + maskv)):
    print("sort by %-32s %.4f of all tiles" % (name, scanned_key(key) / full))
