#!/usr/bin/env python3
"""One-off randomized soak of the training step's fused (two-launch) form under contention: a handful of users and dishes, so that
every wave of a step both numbers rows and waits for numbers other waves assign, every learner, batches of 1 to 1 024 pairs,
loss-only calls and refused steps in between (test infrastructure, not collected by pytest: seeds come from the clock).
Usage on the GPU box: python tests/soak_train_contention.py [cases] [first seed]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from foodrec_amd import ScoringEngine
from oracle import train_oracle as T

dev = lambda a: torch.as_tensor(a, device="cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
print("seed0", seed0)
for it in range(n):
    rng = np.random.default_rng(seed0 + it)
    C = int(rng.choice([4, 4, 3])); E = int(rng.choice([64, 32, 128, 200, 6]))
    U = int(rng.integers(1, 40)); I = int(rng.integers(1, 25))
    learner = str(rng.choice(["adam", "sgd", "adagrad", "rmsprop"])); lr = 0.01
    s = 3.0 / np.sqrt(E)
    PM = (rng.standard_normal((U, C + 1, E)) * s).astype(np.float32)
    RE = (rng.standard_normal((I, E)) * s).astype(np.float32)
    CE = (rng.standard_normal((C, E)) * s).astype(np.float32)
    eng = ScoringEngine(PM.copy(), RE.copy(), CE.copy()); eng.train_begin(learner, lr)
    st = T.TrainState(PM, RE, CE, learner, lr)
    steps = int(rng.integers(2, 6))
    for k in range(steps):
        B = int(rng.choice([int(rng.integers(1, 1025)), 1024, 256, int(rng.integers(1025, 3000))]))
        users = rng.integers(0, U, B).astype(np.int32); items = rng.integers(0, I, B).astype(np.int32)
        cats = rng.integers(0, 2, (B, C)).astype(np.float32)
        cats[cats.sum(1) == 0, rng.integers(0, C)] = 1.0
        labels = rng.integers(0, 2, B).astype(np.float32)
        if rng.integers(0, 4) == 0:                                      # a loss-only call leaves everything as it was
            lo = eng.train_step(dev(users), dev(items), dev(cats), dev(labels), apply=False).cpu().numpy()
            ref_loss, _ = st.step(users, items, cats, labels, apply=False)
            assert abs(lo[0] - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (it, k, "loss only")
        if rng.integers(0, 5) == 0:                                      # a refused step assigns nothing
            bad = users.copy(); bad[rng.integers(0, B)] = U + 3
            eng.train_step(dev(bad), dev(items), dev(cats), dev(labels))
            try:
                eng.check()
                raise AssertionError("a bad id went unnoticed")
            except IndexError:
                pass
        ref_loss, ref_norm = st.step(users, items, cats, labels)
        loss, norm, _, _ = eng.train_step(dev(users), dev(items), dev(cats), dev(labels)).cpu().numpy(); eng.check()
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (it, k, loss, ref_loss)
        assert abs(norm - ref_norm) <= 3e-5 * max(1.0, ref_norm), (it, k, norm, ref_norm)
    tol = 1e-3 * lr * steps if learner in ("adam", "rmsprop") else 2e-5
    for got, ref in ((eng.pm, st.PM), (eng.re, st.RE), (eng.ce, st.CE)):
        err = np.abs(got.cpu().numpy().astype(np.float64) - ref)
        bound = tol * np.maximum(1.0, np.abs(ref)) if learner in ("sgd", "adagrad") else tol
        assert np.all(err <= bound), (it, learner, float(err.max()))
    assert eng.train_steps() == steps
    eng.train_end()
    if it % 10 == 0:
        print("ok", it, learner, "C%d E%d U%d I%d steps %d" % (C, E, U, I, steps), flush=True)
print("all", n, "cases agree")
