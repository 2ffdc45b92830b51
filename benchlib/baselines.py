"""The CPU baselines: the oracle timed on this box's host cores, and the live parity checks of the TIMED kernels.
These functions (and evaluator.py's CPU loop) are the only places the bench imports anything from oracle/."""
from __future__ import annotations

import os
import time

from .common import PARITY_TOL, usable_cores


def mlp_baseline(torch, PM, RE, CE, dish_cats, head, users, items, user_base, gpu_sample, budget_s):
    """The build's float64 restatement of the 3-layer head (oracle/m2d_oracle.py::inference_mlp; the head has no
    reference counterpart) on the first pairs of the timed batch: a live parity check of the TIMED kernel's scores,
    and its rate on this box's host cores (numpy / BLAS threads as configured) beside the GPU number."""
    import numpy as np
    from oracle import m2d_oracle
    n = gpu_sample.numel()
    pm, re, ce, dc = PM.cpu().numpy(), RE.cpu().numpy(), CE.cpu().numpy(), dish_cats.cpu().numpy()
    hd = [h.cpu().numpy() if hasattr(h, "cpu") else h for h in head]
    u = (users[:n].cpu().numpy() - int(user_base)).astype(np.int64)
    d = items[:n].cpu().numpy().astype(np.int64)
    ref = m2d_oracle.inference_mlp(pm, re, ce, dc, *hd, u, d)                 # float64: the parity sample
    # the rate: the same arithmetic in float32 with the dish vectors built once (as the engine keeps them), on
    # slices of 65536 pairs of the timed batch
    Dt = m2d_oracle.dish_vectors(re, ce, dc, m2d_oracle.DEFAULT_COEF, np.float32)
    W1, b1, W2, b2, w3, b3 = [np.asarray(x, dtype=np.float32) for x in hd]
    nb = min(65536, users.numel())
    ub = (users[:nb].cpu().numpy() - int(user_base)).astype(np.int64)
    db = items[:nb].cpu().numpy().astype(np.int64)
    pm2 = pm.reshape(pm.shape[0], -1)
    calls, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < min(budget_s, 10.0) * 0.5:
        z = pm2[ub] * Dt[db]
        h2 = np.maximum(np.maximum(z @ W1 + b1, 0) @ W2 + b2, 0)
        (z.sum(axis=1) + (h2 @ w3 + b3)).sum()
        calls += 1
    rate = calls * nb / (time.perf_counter() - t0)
    got = gpu_sample.cpu().numpy().astype(np.float64)
    err = float(np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
    ok = bool(err <= PARITY_TOL and np.array_equal(np.isnan(got), np.isnan(ref)))
    return ({"value": rate, "unit": "pairs/s", "cores": usable_cores(), "kind": "port",
             "sample": "numpy float32 restatement of the build-defined head (gather, multiply, two BLAS GEMMs, dot) on "
                       "%d-pair slices of the timed batch, dish vectors built once, %d calls; parity: float64 "
                       "restatement on the first %d pairs" % (nb, calls, n),
             "max_rel_diff_vs_gpu": err, "parity_tolerance": PARITY_TOL, "parity_ok": ok}, ok)


def cpu_baseline(torch, PM, RE, CE, users, items, cats, budget_s):
    """CPU restatement of the reference graph (oracle/torch_graph.py) on this box's host cores.

    Times three call sizes of the same workload -- the reference's own 51 pairs per call (evaluate.py:39-58), 4096
    and 65536 -- and reports the fastest as `value`, so the baseline is the most favourable batching of the
    op-for-op graph, not a strawman."""
    from oracle import c_oracle, torch_graph
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    Bc = min(1 << 18, users.numel())
    pm, re, ce = PM.cpu(), RE.cpu(), CE.cpu()
    u, d, m = users[:Bc].cpu(), items[:Bc].cpu(), cats[:Bc].cpu()
    ref = torch_graph.inference(pm, re, ce, u, d, m)                         # also the parity sample
    rates = {}
    for size in (51, 4096, 65536):
        calls, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s * 0.22:
            o = (calls * size) % (Bc - size)
            torch_graph.inference(pm, re, ce, u[o:o + size], d[o:o + size], m[o:o + size])
            calls += 1
        rates[size] = calls * size / (time.perf_counter() - t0)
    best = max(rates, key=rates.get)
    # context: the fused scalar C port of the same formula (no temporaries), all OpenMP threads
    pmn, ren, cen = pm.numpy(), re.numpy(), ce.numpy()
    un, dn, mn = u.numpy(), d.numpy(), m.numpy()
    cthreads = min(ncores, c_oracle.max_threads())
    c_oracle.score_pairs(pmn, ren, cen, un[:4096], dn[:4096], mn[:4096], nthreads=cthreads)
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s * 0.2:
        c_oracle.score_pairs(pmn, ren, cen, un, dn, mn, nthreads=cthreads)
        reps += 1
    c_rate = reps * Bc / (time.perf_counter() - t0)
    return {"value": rates[best], "unit": "pairs/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": "CPU restatement of reference graph (TF unavailable): torch-CPU op-for-op with [B,C,E] "
                      "temporaries on %d-pair slices of the same workload, ~%.0f s per call size; pairs/s at "
                      "51 / 4096 / 65536 pairs per call = %.3g / %.3g / %.3g (value = best, %d per call)"
                      % (Bc, budget_s * 0.22, rates[51], rates[4096], rates[65536], best),
            "value_51_pair_calls": rates[51],
            "host_cpu_count": os.cpu_count(),
            "fused_c_port": {"value": c_rate, "unit": "pairs/s", "cores": cthreads,
                             "what": "oracle/m2d_oracle.c, fused scalar loop, OpenMP"}}, ref, Bc


def check_timed_sample(torch, cb, timed_sample, ref):
    """The baseline doubles as a live parity check of the TIMED kernel's output (sampled right after the timed
    region, before any side leg ran) on the same pairs.  Returns parity_ok."""
    got = timed_sample.cpu()
    err = ((got - ref).abs() / ref.abs().clamp(min=1.0)).max().item()
    cb["max_abs_diff_vs_gpu"] = (got - ref).abs().max().item()
    cb["max_rel_diff_vs_gpu"] = err
    cb["parity_tolerance"] = PARITY_TOL
    cb["parity_ok"] = bool(err <= PARITY_TOL and torch.equal(torch.isnan(got), torch.isnan(ref)))
    return cb["parity_ok"]
