"""Constants, SURVEY.md 8d's byte model, synthetic inputs and HIP-event timing shared by every leg."""
from __future__ import annotations

import os
import time

HBM_PEAK_GBS = 8000.0              # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy rate
INFINITY_CACHE_BYTES = 256 << 20   # tables up to this size can stay in the die-level cache between launches
PARITY_TOL = 1e-4                  # north_star: scores within 1e-4 fp32, as |d| <= tol * max(1, |ref|)
BF16_MFMA_PEAK_TFLOPS = 2500.0     # dense bf16 (no sparsity)
F32_MFMA_PEAK_TFLOPS = 157.3
XGMI_LINK_GBS = 153.0              # one xGMI link, per direction (7 links per GPU)
# What the matrix pipe sustains on this part in a loop of nothing but v_mfma_f32_32x32x16_bf16 from registers, every
# CU, two waves per SIMD (scripts/diag/mfma_chain_probe.cpp, profiles/r05_mfma_chain_probe.txt)
BARE_BF16_MFMA_LOOP = {
    "zero_operands_TFLOPs": 2460.0, "random_operands_TFLOPs": 1865.0,
    "source": "profiles/r05_mfma_chain_probe.txt: 2.38 GHz on zeros, 1.83 GHz on N(0, 1) bf16 operands; `peak` "
              "stays the spec figure (2 500 at 2.4 GHz)"}


def algorithmic_bytes_per_pair(C: int, E: int, mean_active=None):
    """user block + dish row + mask + two ids + score.

    `mean_active` (the batch's mean number of categories with a non-zero mask weight): the byte count of the path
    as built -- the Personal_Memory row of a category whose weight is 0 is multiplied by 0 in the reference graph
    (Model_Recommender.py:82) and is not fetched, so a pair needs U_high + `active` low-level rows."""
    if mean_active is None:
        return (C + 2) * E * 4 + C * 4 + 12
    return (2.0 + mean_active) * E * 4 + C * 4 + 12


def bare_loop_fields(achieved_tflops):
    return {"bare_mfma_loop": BARE_BF16_MFMA_LOOP,
            "frac_of_bare_mfma_loop_random_operands": achieved_tflops / BARE_BF16_MFMA_LOOP["random_operands_TFLOPs"]}


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def masks_from_patterns(torch, pat, C, dev):
    """0/1 category masks f32[n, C] from bit patterns i32[n]."""
    bits = torch.arange(C, device=dev, dtype=torch.int32)[None, :]
    return ((pat[:, None] >> bits) & 1).to(torch.float32).contiguous()


def random_masks(torch, n, C, dev, gen):
    """Uniformly random NON-EMPTY category subsets: (patterns i32[n], masks f32[n, C])."""
    pat = torch.randint(1, 2 ** C, (n,), generator=gen, device=dev, dtype=torch.int32)
    return pat, masks_from_patterns(torch, pat, C, dev)


def make_inputs(torch, dev, U, I, C, E, B, seed, user_base):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    s = 1.0 / (E ** 0.5)
    PM = torch.randn((U, C + 1, E), generator=g, device=dev, dtype=torch.float32) * s
    RE = torch.randn((I, E), generator=g, device=dev, dtype=torch.float32) * s
    CE = torch.randn((C, E), generator=g, device=dev, dtype=torch.float32) * s
    users = torch.randint(0, U, (B,), generator=g, device=dev, dtype=torch.int32) + int(user_base)
    items = torch.randint(0, I, (B,), generator=g, device=dev, dtype=torch.int32)
    _, cats = random_masks(torch, B, C, dev, g)
    return PM, RE, CE, users, items, cats


def time_steps(torch, eng, users, items, cats, out, steps, step=None):
    """K launches; per-launch HIP-event durations (ms) on the current stream + wall seconds."""
    if step is None:
        def step():
            eng.score_pairs(users, items, cats, out=out)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    return wall, per


def settle(torch, step, ms=100.0):
    """Untimed launches of `step` for `ms` of wall time: a leg that follows host work (an idle GPU) would otherwise be
    timed while the part ramps its clocks (bench.py's config.settle does the same in front of the headline's warmup)."""
    t0, n = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        step()
        n += 1
        torch.cuda.synchronize()
    return n


def usable_cores():
    """Host cores this process may actually run on: the affinity mask capped by the cgroup CPU quota.  (A GPU box
    hands one GPU's job a share of a 256-thread host; 256 threads on that share run slower than 16.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            if parse:
                quota, period = parse(open(path).read())
            else:
                quota = open(path).read().strip()
                period = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1"):
                n = min(n, max(1, -(-int(quota) // int(period))))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


class Clock:
    """HIP events on the current stream for a GPU device, perf_counter on CPU (the gloo test of the sharded legs)."""

    def __init__(self, torch, dev):
        self.torch, self.gpu = torch, torch.device(dev).type == "cuda"

    def mark(self):
        if self.gpu:
            e = self.torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        return time.perf_counter()

    def sync(self):
        if self.gpu:
            self.torch.cuda.synchronize()

    def ms(self, a, b):
        return a.elapsed_time(b) if self.gpu else (b - a) * 1e3
