#!/usr/bin/env python3
"""The driver's per-batch call (Train_recommender.py:189-199): sess.run([loss_value, learning_rate, general, train_op], feed)
with Python-list feeds of 128 pairs at the reference's sizes -- wall time per call and where the host time goes."""
import os, sys, time, types, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import foodrec_amd

U, I, C, E, L, B = 64657, 4548, 4, 200, 95, 128
learner = sys.argv[1] if len(sys.argv) > 1 else "adam"
rng = np.random.default_rng(0)
args = types.SimpleNamespace(num_categories=C, num_users=U, embed_size=E, high_level_score_coefficient=0.99, learner=learner, lr=0.001,
                             decay_steps=1000, decay_rate=1.0, num_user_labels=L, beta_1=0.5, beta_2=0.5, alpha=0.001)
PM = (rng.standard_normal((U, C + 1, E)) / 8).astype(np.float32); RE = (rng.standard_normal((I, E)) / 8).astype(np.float32)
CE = (rng.standard_normal((C, E)) / 8).astype(np.float32); GM = (rng.standard_normal((L, C + 1, E)) / 8).astype(np.float32)
model = foodrec_amd.Model(args, PM, RE, CE, GM)
sess = foodrec_amd.Session(model)
def batch():
    items = rng.integers(0, I, B).tolist()
    cats = [[[float(x)] for x in row] for row in rng.integers(0, 2, (B, C))]
    for c in cats:
        if sum(v[0] for v in c) == 0: c[0][0] = 1.0
    y = (rng.random((B, L)) < 0.05).astype(np.float32); y[y.sum(1) == 0, 0] = 1
    return {model.user_input: rng.integers(0, U, B).tolist(), model.item_input: items, model.labels: rng.integers(0, 2, B).tolist(),
            model.categories: cats, model.user_one_hot_label: y.tolist(), model.write_sign: [[1.0]] * B,
            model.dropout_keep_prob: 0.8, model.is_training_flag: True}
feeds = [batch() for _ in range(300)]
fetch = [model.loss_value, model.learning_rate, model.general, model.train_op]
for f in feeds[:20]: sess.run(fetch, f)
torch.cuda.synchronize(); t0 = time.perf_counter()
for f in feeds: sess.run(fetch, f)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / len(feeds)
print("%s: %.1f us per driver-shaped training call (batch %d, Python-list feeds)" % (learner, dt * 1e6, B))
pr = cProfile.Profile(); pr.enable()
for f in feeds[:200]: sess.run(fetch, f)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
