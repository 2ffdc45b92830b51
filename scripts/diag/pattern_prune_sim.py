"""How much of the catalogue a block of 256 users has to scan when mask patterns that cannot reach a user's top-k are left
out (rigorous bounds: every dish of pattern P scores within alpha_P +- |w_P| max|r|), with the users of a call sorted so
that a block's users share their best patterns.  numpy, CPU.   python scripts/diag/pattern_prune_sim.py [users] [dishes] [E] [scale_low]"""
import sys
import numpy as np

NU = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
I = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
E = int(sys.argv[3]) if len(sys.argv) > 3 else 64
scale_low = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0      # low-level rows scaled by this (trained tables may differ)
C, k = 4, 10
rng = np.random.default_rng(0)
s = E ** -0.5
RE = (rng.standard_normal((I, E)) * s).astype(np.float32)
PM = (rng.standard_normal((NU, C + 1, E)) * s).astype(np.float32)
PM[:, 1:] *= scale_low
CE = (rng.standard_normal((C, E)) * s).astype(np.float32)
pat = rng.integers(1, 16, I)
a, b = np.float32(0.99), np.float32(1) - np.float32(0.99)
norm = np.linalg.norm(RE, axis=1)
rmax = np.array([norm[pat == q].max() if (pat == q).sum() >= k else np.inf for q in range(16)])
rows = np.array([(pat == q).sum() for q in range(16)])
hc = PM[:, 0] @ CE.T
aP = np.full((NU, 16), -np.inf, np.float32)
reach = np.zeros((NU, 16), np.float32)
for q in range(1, 16):
    cs = [c for c in range(C) if (q >> c) & 1]
    aP[:, q] = a * hc[:, cs].sum(1) / len(cs)
    reach[:, q] = np.linalg.norm(b * PM[:, 1:][:, cs].sum(1) / len(cs), axis=1) * (rmax[q] if np.isfinite(rmax[q]) else 0)
lo = np.where(rows[None, :] >= k, aP - reach * 1.0001, -np.inf)
L = lo[:, 1:].max(1)                                                   # the user's scan-start bound
rel = (aP + reach * 1.0001 >= L[:, None]) & (rows[None, :] > 0)         # patterns that can reach the top-k
rel[:, 0] = False
print("relevant patterns per user: mean %.2f  (1: %.2f  2: %.2f  3+: %.2f)" %
      (rel.sum(1).mean(), (rel.sum(1) == 1).mean(), (rel.sum(1) == 2).mean(), (rel.sum(1) >= 3).mean()))
mask = (rel * (1 << np.arange(16))[None, :]).sum(1)
best = aP.argmax(1)
sec = np.where(rel.sum(1) >= 2, np.argsort(-np.where(rel, aP, -np.inf), axis=1)[:, 1], 0)
for name, key in (("call order", np.arange(NU)), ("sorted by best pattern", best * NU + np.arange(NU)),
                  ("sorted by (best, second)", (best * 16 + sec) * NU + np.arange(NU)), ("sorted by mask value", mask * NU + np.arange(NU))):
    order = np.argsort(key, kind="stable")
    frac = []
    for b0 in range(0, NU, 256):
        u = order[b0:b0 + 256]
        union = rel[u].any(0)
        frac.append(rows[union].sum() / I)
    print("%-28s dishes a block scans: mean %.3f of the catalogue (max %.3f)" % (name, np.mean(frac), np.max(frac)))

# Dispatch order: (block, dish-range split) items handed to 256 CUs in launch order (x = block fastest, then split) against
# longest-first; an item's cost = the relevant 32-dish tiles inside its dish range + a fixed prologue (in tiles).
import heapq
NS, CUS, FIX = 8, 256, 6
order = np.argsort(mask * NU + np.arange(NU), kind="stable")
po = np.argsort(pat, kind="stable")                                      # sorted table: pattern groups in pattern order
tiles_of = np.zeros(16, np.int64)
start = np.zeros(17, np.int64)
for q in range(1, 16):
    tiles_of[q] = -(-rows[q] // 32)
start[1:] = np.cumsum(tiles_of)
T = start[16]
items = []
for bi, b0 in enumerate(range(0, NU, 256)):
    union = rel[order[b0:b0 + 256]].any(0)
    for sp in range(NS):
        t0, t1 = T * sp // NS, T * (sp + 1) // NS
        w = sum(max(0, min(t1, start[q + 1]) - max(t0, start[q])) for q in range(1, 16) if union[q])
        items.append((sp, bi, w + FIX))
def makespan(seq):
    cu = [0.0] * CUS
    heapq.heapify(cu)
    for w in seq:
        heapq.heappush(cu, heapq.heappop(cu) + w)
    return max(cu)
launch = [w for _, _, w in sorted(items)]
total = sum(launch)
print("items %d  total work %.0f tiles  ideal %.1f per CU  largest item %d" % (len(items), total, total / CUS, max(launch)))
print("launch order (block fastest): makespan %.1f = %.2f x ideal" % (makespan(launch), makespan(launch) / (total / CUS)))
by_block = [w for _, _, w in sorted(items, key=lambda t: (t[1], t[0]))]
print("launch order (split fastest): makespan %.1f = %.2f x ideal" % (makespan(by_block), makespan(by_block) / (total / CUS)))
lpt = sorted(launch, reverse=True)
print("longest first:                makespan %.1f = %.2f x ideal" % (makespan(lpt), makespan(lpt) / (total / CUS)))
