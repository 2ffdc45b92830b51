"""--workload train: single-GPU training-step throughput; not the headline metric."""
from __future__ import annotations

import sys

from .common import HBM_PEAK_GBS, make_inputs, time_steps


def train_workload(a, torch, foodrec_amd, dev):
    """Default shape = the reference's flags (Train_recommender.py:35, :51-58): 64 657 users, 4 548 dishes, E = 200,
    batch 128."""
    def given(name):
        return any(x == name or x.startswith(name + "=") for x in sys.argv)
    U = a.users if given("--users") else 64657
    I = a.dishes if given("--dishes") else 4548
    E = a.embed if given("--embed") else 200
    B = a.pairs if given("--pairs") else 128
    C = 4
    PM, RE, CE, users, items, cats = make_inputs(torch, dev, U, I, C, E, B, 20260101 + 6, 0)
    labels = (torch.rand(B, device=dev) < 0.5).float()
    eng = foodrec_amd.ScoringEngine(PM, RE, CE, coef=0.99, device=dev)
    eng.train_begin(a.learner, 0.001)

    def step():
        eng.train_step(users, items, cats, labels)
    for _ in range(a.warmup):
        step()
    eng.check()
    wall, per = time_steps(torch, eng, users, items, cats, None, a.steps, step)
    eng.check()
    avg_ms = sum(per) / len(per)
    table_bytes = 4 * (PM.numel() + RE.numel() + CE.numel())
    dense = a.learner.lower() == "adam"
    # Adam (TF 1.x, not lazy): var, m, v of EVERY row read and written.  Others: the batch's rows only.
    pair_bytes = (2 * (C + 2) * E * 4 + C * 4 + 12) * B          # forward gather + gradient rows out
    alg = (6 * table_bytes if dense else 0) + pair_bytes
    ach = alg / (avg_ms * 1e-3) / 1e9
    note = ("whole step (claim + grad + reduce + finalize + 3 apply + 2 cleanup launches) over the bytes the update "
            "rule must move: 6 x table bytes for TF 1.x Adam, which decays and moves every row every step" if dense
            else "whole step over the batch rows' bytes; launch-bound at this batch size")
    return {"metric": "trained (user,dish) pairs/sec", "value": B * a.steps / wall, "unit": "pairs/s", "n_gpus": 1,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "training step of Model_Recommender.py:99-104, :223-241 on %d users x %d dishes, "
                                   "C=4, E=%d, batch %d, %s; NOT the headline metric" % (U, I, E, B, a.learner),
                       "notes": "sigmoid-CE loss, gradients, global-norm clip 5.0, the update as TF 1.x applies it "
                                "(SURVEY.md 8f row N4)",
                       "users": U, "dishes": I, "embed_size": E, "batch": B, "learner": a.learner},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": None, "step_avg_ms": avg_ms,
                         "algorithmic_bytes_per_step": alg, "note": note}}
