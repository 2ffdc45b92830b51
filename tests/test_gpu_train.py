"""GPU parity of the training step (m2d_train_begin / m2d_train_step) against the build's
restatement of the TF 1.x rules (oracle/train_oracle.py -- PARITY UNPINNED, see its header).

Tolerances.  Loss and gradient norm: 1e-5 relative.  Tables after k steps: float32 arithmetic against a float64
oracle; Adam divides by sqrt(v) + 1e-8, so an element whose gradient is ~0 in float32 but not in float64 may move
by up to lr in one and not the other -- the bound on a table is therefore stated as a fraction of what the step
moved (1e-3 of lr per step for adam / rmsprop, 1e-5 relative for sgd / adagrad)."""
import os
import types

import numpy as np
import pytest

from helpers import random_case

pytestmark = pytest.mark.gpu


def _batches(U, I, C, B, steps, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(steps):
        users = rng.integers(0, U, B).astype(np.int32)
        items = rng.integers(0, I, B).astype(np.int32)
        users[: B // 4] = users[0]                          # duplicate ids: their rows must be summed
        items[B // 8: B // 2] = items[B // 8]
        cats = rng.integers(0, 2, (B, C)).astype(np.float32)
        cats[cats.sum(1) == 0, rng.integers(0, C)] = 1.0
        cats[B // 3] *= 0.5                                 # masks are weights
        labels = rng.integers(0, 2, B).astype(np.float32)
        out.append((users, items, cats, labels))
    return out


def _engine(PM, RE, CE):
    from foodrec_amd import ScoringEngine
    return ScoringEngine(PM.copy(), RE.copy(), CE.copy())


@pytest.mark.parametrize("learner", ["adam", "sgd", "adagrad", "rmsprop"])
@pytest.mark.parametrize("U,I,C,E,B,form", [(300, 100, 4, 32, 128, 0), (64, 40, 4, 200, 8, 0), (50, 30, 3, 6, 257, 0), (2000, 500, 4, 64, 4096, 0),
                                            (300, 100, 4, 32, 128, 14), (64, 40, 4, 200, 1000, 0), (64, 40, 4, 200, 1000, 14)])
def test_train_steps_match_restatement(learner, U, I, C, E, B, form):
    """Batches of up to 1024 pairs run the two-launch form (m2d_train_grad_fused + m2d_train_apply_fused), larger ones --
    or option "variant" = 14 -- the nine-launch form; same step either way."""
    import torch
    from oracle import train_oracle as T
    PM, RE, CE, *_ = random_case(U, I, C, E, 1, seed=U + E)
    PM, RE, CE = PM * 3, RE * 3, CE * 3
    lr = 0.01
    eng = _engine(PM, RE, CE)
    eng.set_option("variant", form)
    eng.train_begin(learner, lr)
    st = T.TrainState(PM, RE, CE, learner, lr)
    steps = 3
    for k, (users, items, cats, labels) in enumerate(_batches(U, I, C, B, steps, seed=B)):
        ref_loss, ref_norm = st.step(users, items, cats, labels)
        out = eng.train_step(torch.as_tensor(users, device="cuda"), torch.as_tensor(items, device="cuda"),
                             torch.as_tensor(cats, device="cuda"), torch.as_tensor(labels, device="cuda"))
        eng.check()
        loss, norm, scale, got_lr = out.cpu().numpy()
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (k, loss, ref_loss)
        assert abs(norm - ref_norm) <= 1e-5 * max(1.0, ref_norm), (k, norm, ref_norm)
        assert got_lr == np.float32(lr)
        assert scale == pytest.approx(5.0 * min(1.0 / ref_norm, 0.2), rel=1e-5)
    assert eng.last_kernel() == ("m2d_train_grad_fused" if B <= 1024 and form != 14 else "m2d_train_grad")
    tol = 1e-3 * lr * steps if learner in ("adam", "rmsprop") else 1e-5
    for name, got, ref, ini in (("PM", eng.pm, st.PM, PM), ("RE", eng.re, st.RE, RE), ("CE", eng.ce, st.CE, CE)):
        got = got.cpu().numpy().astype(np.float64)
        assert np.abs(ref - ini).max() > 0, name            # the step did something
        err = np.abs(got - ref)
        bound = tol * np.maximum(1.0, np.abs(ref)) if learner in ("sgd", "adagrad") else tol
        assert np.all(err <= bound), "%s %s: max err %.3e" % (learner, name, err.max())
    # optimizer slots (what a checkpoint would hold)
    if learner == "adam":
        for tb, ref_t in enumerate(st.slots):
            for sl in range(2):
                got = eng.train_slot(tb, sl).cpu().numpy()
                np.testing.assert_allclose(got, ref_t[sl], rtol=2e-4, atol=1e-9 if sl else 1e-7)
    elif learner == "sgd":
        with pytest.raises(ValueError):
            eng.train_slot(0, 0)
    eng.train_end()


def test_adam_at_the_reference_default_sizes():
    """Train_recommender.py:35, :51-58: 64 657 users, 4 548 dishes, E = 200, batch 128, adam at lr 0.001 -- two steps
    of the dense (every-row) update against the float32-mode restatement."""
    import torch
    from oracle import train_oracle as T
    U, I, C, E, B = 64657, 4548, 4, 200, 128
    PM, RE, CE, *_ = random_case(U, I, C, E, 1, seed=77)
    PM, RE, CE = PM * 3, RE * 3, CE * 3
    eng = _engine(PM, RE, CE)
    eng.train_begin("adam", 0.001)
    st = T.TrainState(PM, RE, CE, "adam", 0.001, dtype=np.float32)
    for users, items, cats, labels in _batches(U, I, C, B, 2, seed=3):
        ref_loss, _ = st.step(users, items, cats, labels)
        out = eng.train_step(torch.as_tensor(users, device="cuda"), torch.as_tensor(items, device="cuda"),
                             torch.as_tensor(cats, device="cuda"), torch.as_tensor(labels, device="cuda")).cpu().numpy()
        eng.check()
        assert abs(out[0] - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
    for got, ref, ini in ((eng.pm, st.PM, PM), (eng.re, st.RE, RE), (eng.ce, st.CE, CE)):
        got = got.cpu().numpy()
        moved = np.abs(ref - ini) > 0
        assert moved.any() and not moved.all() or ref.size == C * E       # adam moved the touched rows only so far
        assert np.abs(got - ref).max() <= 2e-6                           # 2 steps of lr 0.001; see the module docstring
        assert np.array_equal(got[~moved], ini[~moved])                  # rows without history are bit-for-bit untouched


def test_clip_engages_and_loss_only_leaves_tables_alone():
    import torch
    from oracle import train_oracle as T
    U, I, C, E, B = 200, 80, 4, 64, 512
    PM, RE, CE, *_ = random_case(U, I, C, E, 1, seed=11)
    PM, RE, CE = PM * 200, RE * 200, CE * 200               # huge logits -> gradient norm well above 5
    (users, items, cats, labels), = _batches(U, I, C, B, 1, seed=5)
    eng = _engine(PM, RE, CE)
    eng.train_begin("sgd", 0.5)
    dev = lambda a: torch.as_tensor(a, device="cuda")
    out = eng.train_step(dev(users), dev(items), dev(cats), dev(labels), apply=False).cpu().numpy()
    eng.check()
    st = T.TrainState(PM, RE, CE, "sgd", 0.5)
    ref_loss, ref_norm = st.step(users, items, cats, labels, apply=False)
    assert ref_norm > 5.0
    assert abs(out[0] - ref_loss) <= 1e-5 * abs(ref_loss) and abs(out[1] - ref_norm) <= 1e-5 * ref_norm
    assert torch.equal(eng.pm.cpu(), torch.as_tensor(PM)) and torch.equal(eng.re.cpu(), torch.as_tensor(RE))
    assert torch.equal(eng.ce.cpu(), torch.as_tensor(CE))
    out = eng.train_step(dev(users), dev(items), dev(cats), dev(labels)).cpu().numpy()
    eng.check()
    st.step(users, items, cats, labels)
    assert out[2] == pytest.approx(5.0 / ref_norm, rel=1e-5)
    for got, ref in ((eng.pm, st.PM), (eng.re, st.RE), (eng.ce, st.CE)):
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-4)
    # scoring after training reads the updated tables
    from oracle import m2d_oracle as oracle
    got = eng.score_pairs(dev(users), dev(items), dev(cats)).cpu().numpy(); eng.check()
    ref = oracle.inference_f64(st.PM.astype(np.float32), st.RE.astype(np.float32), st.CE.astype(np.float32), users, items, cats)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-2)


def test_errors_and_slot_restore():
    import torch
    U, I, C, E, B = 100, 50, 4, 32, 64
    PM, RE, CE, *_ = random_case(U, I, C, E, 1, seed=3)
    (users, items, cats, labels), = _batches(U, I, C, B, 1, seed=1)
    dev = lambda a: torch.as_tensor(a, device="cuda")
    eng = _engine(PM, RE, CE)
    with pytest.raises(ValueError, match="m2d_train_begin"):
        eng.train_step(dev(users), dev(items), dev(cats), dev(labels))
    eng.train_begin("adam", 0.001)
    bad = users.copy(); bad[5] = U + 3
    with pytest.raises(IndexError, match="user id %d at position 5" % (U + 3)):
        eng.train_step(dev(bad), dev(items), dev(cats), dev(labels)); eng.check()
    assert torch.equal(eng.pm.cpu(), torch.as_tensor(PM)) and torch.equal(eng.re.cpu(), torch.as_tensor(RE))   # nothing applied
    assert float(eng.train_slot(0, 0).abs().max()) == 0.0
    # ... the step count and Adam's beta powers included (TF raises before the beta-power assigns): the engine goes on
    # WITHOUT a reset exactly like one that never saw the bad batch
    assert eng.train_steps() == 0
    # resume: a second engine fed the first one's tables and slots continues identically
    eng2 = _engine(PM, RE, CE)
    eng2.train_begin("adam", 0.001)
    eng.train_step(dev(users), dev(items), dev(cats), dev(labels)); eng.check()
    eng2.train_step(dev(users), dev(items), dev(cats), dev(labels)); eng2.check()
    for tb in range(3):
        for sl in range(2):
            np.testing.assert_allclose(eng.train_slot(tb, sl).cpu().numpy(), eng2.train_slot(tb, sl).cpu().numpy(), rtol=1e-5, atol=1e-9)
            eng2.train_slot(tb, sl, restore=eng.train_slot(tb, sl))
            assert torch.equal(eng.train_slot(tb, sl), eng2.train_slot(tb, sl))
    assert eng.train_steps() == 1 and eng2.train_steps() == 1
    for a, b in ((eng.pm, eng2.pm), (eng.re, eng2.re), (eng.ce, eng2.ce)):      # float atomics: equal to the last bits only
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-8)
    snap = eng.pm.clone()
    # a raw-ABI caller that does not check: the latch blocks the following steps too, and none of them counts
    eng.train_step(dev(bad), dev(items), dev(cats), dev(labels))
    eng.train_step(dev(users), dev(items), dev(cats), dev(labels))
    with pytest.raises(IndexError):
        eng.check()
    assert eng.train_steps() == 1 and torch.equal(eng.pm, snap)
    # a pending id error is reported as such by m2d_set_ingredients, not as a malformed CSR
    eng.train_end()
    eng.score_pairs(dev(bad), dev(items), dev(cats))
    with pytest.raises(IndexError, match="user id"):
        eng.set_ingredients(np.zeros((3, E), np.float32), np.zeros(I + 1, np.int32), np.zeros(0, np.int32))


def test_session_serves_the_training_fetches():
    """The two sess.run call shapes of Train_recommender.py:170-199."""
    from foodrec_amd import Model, Session
    from oracle import m2d_oracle as oracle
    from oracle import train_oracle as T
    U, I, C, E, L, B = 120, 60, 4, 32, 7, 16
    PM, RE, CE, *_ = random_case(U, I, C, E, 1, seed=21)
    rng = np.random.default_rng(0)
    GM = (rng.standard_normal((L, C + 1, E)) / 8).astype(np.float32)
    args = types.SimpleNamespace(learner="adam", num_categories=C, num_users=U, num_labels=L, embed_size=E, lr=0.002,
                                 decay_steps=1000, decay_rate=1.0, high_level_score_coefficient=0.99, beta_1=0.01,
                                 beta_2=0.01, alpha=0.01)
    model = Model(args, PM.copy(), RE.copy(), CE.copy(), GM.copy())
    sess = Session(model)
    (users, items, cats, labels), = _batches(U, I, C, B, 1, seed=9)
    sign = np.where(labels > 0, 1.0, -1.0).astype(np.float32)[:, None]
    onehot = (rng.random((B, L)) < 0.3).astype(np.float32); onehot[:, 0] = 1
    feed = {model.user_input: [str(u) for u in users], model.item_input: list(items), model.labels: list(labels),
            model.categories: cats[:, :, None].tolist(), model.user_one_hot_label: onehot.tolist(),
            model.write_sign: sign.tolist(), model.dropout_keep_prob: 0.8, model.is_training_flag: True}
    st = T.TrainState(PM, RE, CE, "adam", 0.002)
    ref_loss0, _ = st.step(users, items, cats, labels, apply=False)
    assert sess.run(model.loss_value, feed) == pytest.approx(ref_loss0, rel=1e-5)          # loss only: no update
    curr_loss, lr, general, _ = sess.run([model.loss_value, model.learning_rate, model.general, model.train_op], feed)
    ref_loss, _ = st.step(users, items, cats, labels)
    assert curr_loss == pytest.approx(ref_loss, rel=1e-5) and lr == np.float32(0.002)
    # the driver's ordinary batch (Train_recommender.py:195-199) fetches `general` only: the General_Memory assign
    # (Model_Recommender.py:215) runs -- after the optimizer step, on the updated tables -- and the two
    # Personal_Memory assigns (:167, :198) do NOT: Personal_Memory is exactly what the optimizer left
    pm32, re32, ce32 = (t.astype(np.float32) for t in (st.PM, st.RE, st.CE))
    refPM, refGM, _, _ = oracle.write_memory(pm32, re32, ce32, GM, users, items, cats, sign, onehot, 0.01, 0.01, 0.01,
                                             personal=False)
    assert np.array_equal(refPM, pm32)                  # personal=False: the oracle leaves Personal_Memory alone
    assert general == pytest.approx(refGM.mean(), rel=1e-4, abs=1e-7)
    np.testing.assert_allclose(model.general_memory(), refGM, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(model.engine.pm.cpu().numpy(), pm32, rtol=1e-3, atol=2e-6)
    written, _, _, _ = oracle.write_memory(pm32, re32, ce32, GM, users, items, cats, sign, onehot, 0.01, 0.01, 0.01)
    assert np.abs(written - pm32).max() > 1e-4          # the write would have been visible at this tolerance
    # `personal` alone: both Personal_Memory assigns, General_Memory untouched (g_b reads it as it stands)
    pm_before, gm_before = model.engine.pm.cpu().numpy().copy(), model.general_memory().copy()
    personal = sess.run(model.personal, feed)
    refPM2, refGM2, mp, _ = oracle.write_memory(pm_before, re32, ce32, gm_before, users, items, cats, sign, onehot,
                                                0.01, 0.01, 0.01, general=False)
    assert personal == pytest.approx(mp, rel=1e-4, abs=1e-7)
    np.testing.assert_allclose(model.engine.pm.cpu().numpy(), refPM2, rtol=1e-4, atol=2e-6)
    np.testing.assert_array_equal(model.general_memory(), gm_before)
    # the rare 'Write Personal Memory' branch (:180-184) fetches both
    loss2, lr2, personal, general2, _ = sess.run([model.loss_value, model.learning_rate, model.personal, model.general,
                                                  model.train_op], feed)
    assert np.isfinite(personal) and np.isfinite(general2) and loss2 < curr_loss + 1.0
    assert sess.run(model.epoch_increment) == 1
    with pytest.raises(ValueError, match="labels"):
        sess.run([model.train_op], {k: v for k, v in feed.items() if k is not model.labels})


def test_driver_shaped_loop_end_to_end(tmp_path):
    """The inner loop of Train_recommender.py:124-216 on files in the reference's formats: load, build the training
    feeds, run batches through sess.run([loss_value, learning_rate, general, train_op]) (the first batch through the
    8-pair memory-write branch, :169-186), evaluate.  Checks the plumbing and that training lowers the training loss;
    the numbers themselves are pinned by the tests above."""
    import random
    from foodrec_amd import Model, Session, evaluate_model, formats
    base = formats.write_synthetic_split(str(tmp_path), num_users=60, num_dishes=50, embed_size=32, num_labels=6,
                                         train_per_user=4)
    ds = formats.Dataset(base)
    load = lambda n: formats.load_numpy_file(os.path.join(str(tmp_path), n))
    d2c = formats.load_json_file(os.path.join(str(tmp_path), "dish_to_category.json"))
    u2l = formats.load_json_file(os.path.join(str(tmp_path), "user_to_one_hot_label.json"))
    args = types.SimpleNamespace(learner="adam", num_categories=4, num_users=60, num_labels=6, embed_size=32, lr=0.01,
                                 decay_steps=1000, decay_rate=1.0, high_level_score_coefficient=0.99, beta_1=0.01,
                                 beta_2=0.01, alpha=0.01)
    model = Model(args, load("Personal_Memory.npy"), load("Recipe_Embedding.npy"), load("Category_Embedding.npy"),
                  load("General_Memory.npy"))
    sess = Session(model)
    random.seed(1)
    users, items, labels, cats, sign, onehot = formats.get_train_instances(ds.trainMatrix, ds.testNegatives, d2c, u2l)
    n, bs = len(users), 128
    assert n == 60 * (4 + 50)
    hits0, _ = evaluate_model(sess, model, ds.testRatings, ds.testNegatives, 10, d2c)
    losses = []
    for epoch in range(3):
        tot = 0.0
        for start in range(0, n - bs + 1, bs):
            sl = slice(start, start + bs)
            feed = {model.user_input: users[sl], model.item_input: items[sl], model.labels: labels[sl],
                    model.categories: cats[sl], model.user_one_hot_label: onehot[sl], model.write_sign: sign[sl],
                    model.dropout_keep_prob: 0.8, model.is_training_flag: True}
            if epoch == 0 and start == 0:                   # the memory-write branch: 16 mini-batches of 8
                for mini in range(16):
                    ms = slice(start + 8 * mini, start + 8 * (mini + 1))
                    mfeed = {k: (v[ms] if isinstance(v, list) else v) for k, v in
                             ((model.user_input, users), (model.item_input, items), (model.labels, labels),
                              (model.categories, cats), (model.user_one_hot_label, onehot), (model.write_sign, sign))}
                    curr_loss, lr, personal, general, _ = sess.run(
                        [model.loss_value, model.learning_rate, model.personal, model.general, model.train_op], mfeed)
            else:
                curr_loss, lr, general, _ = sess.run([model.loss_value, model.learning_rate, model.general, model.train_op], feed)
            assert np.isfinite(curr_loss) and lr == np.float32(0.01)
            tot += float(curr_loss)
        sess.run(model.epoch_increment)
        losses.append(tot)
    assert losses[-1] < losses[0]
    hits, ndcgs = evaluate_model(sess, model, ds.testRatings, ds.testNegatives, 10, d2c)
    assert len(hits) == len(ndcgs) == 60 and len(hits0) == 60
    assert np.isfinite(model.engine.pm.cpu().numpy()).all() and np.isfinite(model.general_memory()).all()


def test_checkpoint_save_restore_resumes_training(tmp_path):
    """Model.save / Model.restore stand where the driver uses tf.train.Saver: tables, optimizer slots and step count
    come back, training continues as if uninterrupted, and retrieval sees the restored tables."""
    import torch
    from foodrec_amd import Model
    U, I, C, E, B = 150, 90, 4, 64, 64
    PM, RE, CE, *_ = random_case(U, I, C, E, 1, seed=31)
    GM = np.zeros((5, C + 1, E), np.float32)
    args = types.SimpleNamespace(learner="adam", num_categories=C, num_users=U, num_labels=5, embed_size=E, lr=0.01,
                                 high_level_score_coefficient=0.99, beta_1=0.01, beta_2=0.01, alpha=0.01)
    batches = _batches(U, I, C, B, 3, seed=8)
    a = Model(args, PM.copy(), RE.copy(), CE.copy(), GM.copy())
    for b in batches[:2]:
        a.train_step(*b)
    ck = os.path.join(str(tmp_path), "ck")
    a.save(ck)
    z = np.load(ck + ".npz")
    assert z["Personal_Memory"].shape == PM.shape and int(z["steps"]) == 2 and str(z["learner"]) == "adam"
    b_model = Model(args, PM.copy(), RE.copy(), CE.copy(), GM.copy())
    dish_cats = np.ones((I, C), np.float32)
    b_model.set_dish_categories(dish_cats)
    before = b_model.topk(np.arange(8), 5)[1]              # builds the retrieval tables from the UNtrained RE
    b_model.restore(ck)
    assert torch.equal(b_model.engine.pm, a.engine.pm) and torch.equal(b_model.engine.re, a.engine.re)
    assert b_model.engine.train_steps() == 2
    la, _ = a.train_step(*batches[2])
    lb, _ = b_model.train_step(*batches[2])
    assert la == pytest.approx(lb, rel=1e-6)
    for x, y in ((a.engine.pm, b_model.engine.pm), (a.engine.re, b_model.engine.re), (a.engine.ce, b_model.engine.ce)):
        assert (x - y).abs().max().item() <= 1e-6           # float-atomic order is the only difference
    a.set_dish_categories(dish_cats)
    sa, ia = a.topk(np.arange(8), 5)
    sb, ib = b_model.topk(np.arange(8), 5)                  # stale tables would still rank with the untrained RE
    np.testing.assert_allclose(sa, sb, rtol=1e-4, atol=1e-5)
    wrong = types.SimpleNamespace(**{**vars(args), "learner": "sgd"})
    with pytest.raises(ValueError, match="trained with adam"):
        Model(wrong, PM.copy(), RE.copy(), CE.copy(), GM.copy()).restore(ck)


def test_the_two_forms_of_a_step_can_alternate():
    """The two-launch form (small batches) and the nine-launch form share the slot maps, the compact gradient rows, the dense
    gradient and the optimizer state: steps of both kinds, loss-only calls and a refused step in between end where the
    restatement ends."""
    import torch
    from oracle import train_oracle as T
    U, I, C, E = 500, 200, 4, 64
    PM, RE, CE, *_ = random_case(U, I, C, E, 1, seed=77)
    PM, RE, CE = PM * 3, RE * 3, CE * 3
    eng = _engine(PM, RE, CE)
    eng.train_begin("adam", 0.01)
    st = T.TrainState(PM, RE, CE, "adam", 0.01)
    dev = lambda x: torch.as_tensor(x, device="cuda")
    sizes = [128, 2048, 64, 64, 1500, 256, 1024, 1025, 8]
    for k, B in enumerate(sizes):
        users, items, cats, labels = next(iter(_batches(U, I, C, B, 1, seed=100 + k)))
        if k in (2, 5):                                      # a loss-only call before the step: leaves everything as it was
            lo = eng.train_step(dev(users), dev(items), dev(cats), dev(labels), apply=False).cpu().numpy()
            ref_loss, _ = st.step(users, items, cats, labels, apply=False)
            assert abs(lo[0] - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
        if k == 4:                                           # a refused step (an id out of range) assigns nothing, in either form
            for Bb in (64, 2000):
                bu, bi, bc, bl = next(iter(_batches(U, I, C, Bb, 1, seed=300 + Bb)))
                bu = bu.copy(); bu[Bb // 2] = U + 5
                eng.train_step(dev(bu), dev(bi), dev(bc), dev(bl))
                with pytest.raises(IndexError):
                    eng.check()
        ref_loss, ref_norm = st.step(users, items, cats, labels)
        out = eng.train_step(dev(users), dev(items), dev(cats), dev(labels)).cpu().numpy(); eng.check()
        assert abs(out[0] - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (k, B)
        assert abs(out[1] - ref_norm) <= 1e-5 * max(1.0, ref_norm), (k, B)
    assert eng.train_steps() == len(sizes)
    for got, ref in ((eng.pm, st.PM), (eng.re, st.RE), (eng.ce, st.CE)):
        assert np.abs(got.cpu().numpy() - ref).max() <= 1e-3 * 0.01 * len(sizes)


@pytest.mark.timeout(120)
def test_crossing_claims_do_not_wait_for_each_other():
    """Small batches: a pair's wave numbers the rows it is first to meet and waits for the number of a row another wave met
    first.  With a handful of users and dishes every wave both numbers and waits, in every combination -- wave A holding
    the user wave B waits for while B holds the dish A waits for is the case that must not hang (a wave publishes its numbers
    before it waits for anybody's).  300 steps at the fused form's largest batch, then the tables against the restatement."""
    import torch
    from oracle import train_oracle as T
    U, I, C, E, B = 5, 4, 4, 64, 1024
    PM, RE, CE, *_ = random_case(U, I, C, E, 1, seed=5)
    eng = _engine(PM, RE, CE)
    eng.train_begin("sgd", 0.05)
    st = T.TrainState(PM, RE, CE, "sgd", 0.05)
    dev = lambda x: torch.as_tensor(x, device="cuda")
    rng = np.random.default_rng(3)
    for step in range(300):
        users = rng.integers(0, U, B).astype(np.int32); items = rng.integers(0, I, B).astype(np.int32)
        cats = rng.integers(0, 2, (B, C)).astype(np.float32); cats[cats.sum(1) == 0, 0] = 1.0
        labels = rng.integers(0, 2, B).astype(np.float32)
        out = eng.train_step(dev(users), dev(items), dev(cats), dev(labels))
        if step < 3:
            ref_loss, ref_norm = st.step(users, items, cats, labels)
            o = out.cpu().numpy(); eng.check()
            assert abs(o[0] - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)) and abs(o[1] - ref_norm) <= 2e-5 * max(1.0, ref_norm), step
            if step == 2:
                for got, ref in ((eng.pm, st.PM), (eng.re, st.RE), (eng.ce, st.CE)):
                    assert np.abs(got.cpu().numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    torch.cuda.synchronize(); eng.check()
    assert eng.train_steps() == 300
