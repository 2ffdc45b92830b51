"""where the cycles of a pruned scan launch go.  Reads a dump of topk_diag (-DM2D_DIAG=8208, env
M2D_DIAG_DUMP=<file>): one 64-byte record per (workgroup, wave) of the LAST scan launch --
  [body cycles, inserts | bx << 20 | by << 52, stage-wait + barrier cycles, candidate-path cycles, candidate steps, steps,
   s_memtime ticks, (start at 100 MHz) << 24 | duration at 100 MHz]
and prints (i) the share of wave cycles in plain bodies / candidate handling / stage waits / the rest, (ii) the launch's time
line: when workgroups start and end, how full the chip's workgroup slots are, and the tail = last end - median end.
    python scripts/diag/scan_breakdown.py <dump> <waves per block> [slots per CU] [CUs]"""
import sys

import numpy as np

path, waves = sys.argv[1], int(sys.argv[2])
slots_per_cu = int(sys.argv[3]) if len(sys.argv) > 3 else (2 if waves == 4 else 1)
cus = int(sys.argv[4]) if len(sys.argv) > 4 else 256
r = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
r = r[: len(r) // waves * waves].reshape(-1, waves, 8)
live = r[:, 0, 6] != 0
r = r[live]
nb = len(r)
body, bar, slow = r[..., 0].astype(float), r[..., 2].astype(float), r[..., 3].astype(float)
nslow, nstep, ticks = r[..., 4].astype(float), r[..., 5].astype(float), r[..., 6].astype(float)
start = (r[..., 7] >> np.uint64(24)).astype(np.int64)
dur = (r[..., 7] & np.uint64(0xFFFFFF)).astype(np.int64)
t0 = start.min()
s_us, e_us = (start - t0) / 100.0, (start - t0 + dur) / 100.0
span = e_us.max()
tot = ticks.sum()
print("workgroups %d x %d waves; launch span %.1f us (first wave start -> last wave end)" % (nb, waves, span))
print("share of all wave cycles: plain + candidate-step bodies %.3f | candidate handling %.3f | stage wait + barrier %.3f | "
      "prologue / pattern switches / publish %.3f" % (body.sum() / tot, slow.sum() / tot, bar.sum() / tot,
                                                     1 - (body.sum() + slow.sum() + bar.sum()) / tot))
print("steps per wave: mean %.1f, median %.0f, p90 %.0f, p99 %.0f, max %.0f; candidate steps %.3f of steps, %.0f cycles each"
      % (nstep.mean(), np.median(nstep), np.percentile(nstep, 90), np.percentile(nstep, 99), nstep.max(),
         nslow.sum() / max(nstep.sum(), 1), slow.sum() / max(nslow.sum(), 1)))
clk = ticks.sum() / (dur.sum() / 100.0) / 1e3
print("in-kernel clock %.3f GHz; cycles per step (all in) mean %.0f" % (clk, tot / max(nstep.sum(), 1)))
bs, be = s_us.min(axis=1), e_us.max(axis=1)               # per workgroup
bd = be - bs
order = np.argsort(-bd)
print("workgroup duration us: mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f; the longest: %s"
      % (bd.mean(), np.median(bd), np.percentile(bd, 90), np.percentile(bd, 99), bd.max(),
         ", ".join("%.0f us (%d steps, start %.0f)" % (bd[i], nstep[i].max(), bs[i]) for i in order[:5])))
slots = slots_per_cu * cus
busy = bd.sum() / (slots * span)
even = bd.sum() / slots
print("workgroup slots %d: busy %.3f of slot-time over the span; even share %.1f us; span / even share = %.2f"
      % (slots, busy, even, span / even))
med_end, last_end = np.median(be), be.max()
print("tail: last end %.1f us - median end %.1f us = %.1f us = %.3f of the span" % (last_end, med_end, last_end - med_end,
                                                                                  (last_end - med_end) / span))
# occupancy over time: workgroups resident in each 10 % of the span
edges = np.linspace(0, span, 11)
occ = [((bs < edges[i + 1]) & (be > edges[i])).sum() for i in range(10)]
print("workgroups resident at some time in each tenth of the span:", occ)
act = [np.clip(np.minimum(be, edges[i + 1]) - np.maximum(bs, edges[i]), 0, None).sum() / (edges[i + 1] - edges[i]) / slots
       for i in range(10)]
print("slot occupancy per tenth of the span:", " ".join("%.2f" % a for a in act))
