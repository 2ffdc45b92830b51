#!/usr/bin/env python3
"""Latency of the reference-shaped call: sess.run([model.logits], feed_dict) with 51 pairs (evaluate.py:55-59)."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import foodrec_amd

U, I, C, E = 64657, 4548, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(0)
PM = (rng.standard_normal((U, C + 1, E)) / 8).astype(np.float32)
RE = (rng.standard_normal((I, E)) / 8).astype(np.float32)
CE = (rng.standard_normal((C, E)) / 8).astype(np.float32)
args = types.SimpleNamespace(num_categories=C, num_users=U, embed_size=E, high_level_score_coefficient=0.99)
model = foodrec_amd.Model(args, PM, RE, CE, None)
sess = foodrec_amd.Session(model)
if len(sys.argv) > 2:
    model.engine.set_option("host_zero_copy", int(sys.argv[2]))
d2c = {str(i): [[float(x)] for x in rng.integers(0, 2, C)] for i in range(I)}
for k in d2c:
    if sum(v[0] for v in d2c[k]) == 0:
        d2c[k][0][0] = 1.0
def feed(u):
    items = rng.integers(0, I, 51).tolist()
    return {model.user_input: [str(u)] * 51, model.item_input: items, model.labels: [0] * 51,
            model.categories: [d2c[str(i)] for i in items], model.dropout_keep_prob: 1.0, model.is_training_flag: False}
feeds = [feed(u) for u in range(2000)]
for f in feeds[:50]:
    sess.run([model.logits], f)
t0 = time.perf_counter()
for f in feeds:
    sess.run([model.logits], f)
dt = time.perf_counter() - t0
print("E=%d: %.1f us per 51-pair sess.run call (%.0f calls/s, %.2f M pairs/s)" % (E, dt / len(feeds) * 1e6, len(feeds) / dt, 51 * len(feeds) / dt / 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for f in feeds[:500]:
    sess.run([model.logits], f)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
