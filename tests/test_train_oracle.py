"""CPU checks of the training-step restatement (oracle/train_oracle.py).

TensorFlow is not in the image and the reference has no training fixture, so the restatement is PARITY UNPINNED.
What can be pinned on CPU: the analytic gradients against autograd over the op-for-op torch graph
(oracle/torch_graph.py), the sigmoid-CE formula against torch's, and the update rules' fixed points."""
import numpy as np
import pytest
import torch

from helpers import random_case
from oracle import torch_graph
from oracle import train_oracle as T


def _batch(seed, U=40, I=30, C=4, E=12, B=64):
    PM, RE, CE, users, items, cats = random_case(U, I, C, E, B, seed, zero_rows=False)
    rng = np.random.default_rng(seed + 1)
    cats = cats * rng.choice([1.0, 0.5, 2.0], size=cats.shape).astype(np.float32)      # masks are weights, not only 0/1
    cats[cats.sum(1) == 0, 1] = 1.0
    labels = rng.integers(0, 2, B).astype(np.float32)
    users[:8] = users[0]                                                               # duplicate ids in the batch
    items[4:12] = items[4]
    return PM * 4, RE * 4, CE * 4, users, items, cats, labels


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_gradients_match_autograd_over_the_graph(seed):
    PM, RE, CE, users, items, cats, labels = _batch(seed)
    s, loss, dUM, dIt, dCE = T.loss_and_gradients(PM, RE, CE, users, items, cats, labels)
    tp, tr, tc = (torch.tensor(t, dtype=torch.float64, requires_grad=True) for t in (PM, RE, CE))
    logits = torch_graph.inference(tp, tr, tc, torch.tensor(users), torch.tensor(items), torch.tensor(cats, dtype=torch.float64))
    tl = torch.nn.functional.binary_cross_entropy_with_logits(logits, torch.tensor(labels, dtype=torch.float64))
    tl.backward()
    assert abs(loss - tl.item()) < 1e-12
    # torch_graph blends with float32 coefficients, as the reference does; so does the oracle
    np.testing.assert_allclose(s, logits.detach().numpy(), rtol=1e-12, atol=1e-12)
    gPM = np.zeros_like(PM, dtype=np.float64); np.add.at(gPM, users, dUM)
    gRE = np.zeros_like(RE, dtype=np.float64); np.add.at(gRE, items, dIt)
    np.testing.assert_allclose(gPM, tp.grad.numpy(), rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(gRE, tr.grad.numpy(), rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(dCE, tc.grad.numpy(), rtol=1e-10, atol=1e-14)


def test_learner_names_follow_the_reference_switch():
    assert T.learner_code("Adam") == T.ADAM and T.learner_code("adagrad") == T.ADAGRAD
    assert T.learner_code("RMSProp") == T.RMSPROP
    assert T.learner_code("sgd") == T.SGD and T.learner_code("momentum") == T.SGD      # the else branch (:234-235)


def test_clip_uses_per_pair_rows_and_leaves_small_gradients_alone():
    PM, RE, CE, users, items, cats, labels = _batch(3)
    st = T.TrainState(PM, RE, CE, "sgd", lr=0.5)
    _, _, dUM, dIt, dCE = T.loss_and_gradients(PM, RE, CE, users, items, cats, labels)
    loss, norm = st.step(users, items, cats, labels)
    assert norm == pytest.approx(np.sqrt((dUM ** 2).sum() + (dIt ** 2).sum() + (dCE ** 2).sum()))
    assert norm < 5.0                                       # scale = 5 * min(1/norm, 1/5) = 1: plain SGD
    gRE = np.zeros_like(RE, dtype=np.float64); np.add.at(gRE, items, dIt)
    np.testing.assert_allclose(st.RE, RE - 0.5 * gRE, rtol=1e-12, atol=1e-15)
    # blow the gradient up: the update is rescaled to global norm 5 over the per-pair rows
    big = T.TrainState(PM * 300, RE * 300, CE * 300, "sgd", lr=1.0)
    _, _, dUM, dIt, dCE = T.loss_and_gradients(big.PM, big.RE, big.CE, users, items, cats, labels)
    ce0 = big.CE.copy()
    _, norm = big.step(users, items, cats, labels)
    assert norm > 5.0
    np.testing.assert_allclose(big.CE, ce0 - dCE * (5.0 / norm), rtol=1e-10, atol=1e-12)


def test_adam_moves_every_row_and_first_step_is_lr_sized():
    PM, RE, CE, users, items, cats, labels = _batch(4)
    st = T.TrainState(PM, RE, CE, "adam", lr=0.001)
    st.step(users, items, cats, labels)
    touched = np.zeros(len(PM), bool); touched[users] = True
    d = np.abs(st.PM - PM)
    assert np.all(d[~touched] == 0)                         # m = v = 0 there: 0 / (0 + eps)
    hit = d[touched][:, 0, :]                               # the high-level row always has a gradient
    assert np.all(hit > 0.0009) and np.all(hit < 0.0011)    # lr_t m / (sqrt(v) + eps) ~ lr at t = 1
    # second step: rows untouched now, but with history, still move (TF 1.x Adam's sparse path is dense)
    before = st.PM.copy()
    other = np.resize(np.setdiff1d(np.arange(len(PM)), users), len(users)).astype(np.int32)
    st.step(other, items, cats, labels)
    only_first = touched.copy(); only_first[other] = False
    assert np.all(np.abs(st.PM - before)[only_first][:, 0, :] > 0)


@pytest.mark.parametrize("learner", ["adagrad", "rmsprop"])
def test_sparse_learners_touch_only_the_batch_rows(learner):
    PM, RE, CE, users, items, cats, labels = _batch(5)
    st = T.TrainState(PM, RE, CE, learner, lr=0.01)
    st.step(users, items, cats, labels)
    touched = np.zeros(len(RE), bool); touched[items] = True
    assert np.all(st.RE[~touched] == RE[~touched])
    assert np.all(np.abs(st.RE - RE)[touched].max(1) > 0)
    if learner == "adagrad":
        assert np.all(st.slots[1][0][~touched] == np.float64(0.1))
    else:
        assert np.all(st.slots[1][0][~touched] == 1.0)      # rms slot starts at one


def test_float32_mode_tracks_float64():
    PM, RE, CE, users, items, cats, labels = _batch(6)
    a, b = T.TrainState(PM, RE, CE, "adam"), T.TrainState(PM, RE, CE, "adam", dtype=np.float32)
    for _ in range(3):
        la, _ = a.step(users, items, cats, labels)
        lb, _ = b.step(users, items, cats, labels)
        assert abs(la - lb) < 1e-5
    assert np.abs(a.PM - b.PM).max() < 2e-5


@pytest.mark.parametrize("learner", ["sgd", "adagrad"])
def test_two_steps_match_torch_optim_where_the_rules_coincide(learner):
    """torch.optim.SGD and torch.optim.Adagrad(initial_accumulator_value=0.1, eps=0) apply the same formulas as the
    TF 1.x optimizers for these two learners (dense autograd gradients sum duplicate rows, as TF's
    _apply_sparse_duplicate_indices does; rows with a zero gradient do not move under either rule).  Adam and
    RMSProp differ between the libraries (epsilon placement, slot initial values) and are not compared."""
    PM, RE, CE, users, items, cats, labels = _batch(7)
    lr = 0.05
    st = T.TrainState(PM, RE, CE, learner, lr=lr)
    tp, tr, tc = (torch.tensor(t, dtype=torch.float64, requires_grad=True) for t in (PM, RE, CE))
    opt = (torch.optim.SGD([tp, tr, tc], lr=float(np.float32(lr))) if learner == "sgd" else
           torch.optim.Adagrad([tp, tr, tc], lr=float(np.float32(lr)), initial_accumulator_value=0.1, eps=0.0))
    for step in range(2):
        u = np.roll(users, step * 3); d = np.roll(items, step * 5)
        ref_loss, norm = st.step(u, d, cats, labels)
        assert norm < 5.0                                   # no clipping in this case: torch has none here
        opt.zero_grad()
        logits = torch_graph.inference(tp, tr, tc, torch.tensor(u), torch.tensor(d), torch.tensor(cats, dtype=torch.float64))
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, torch.tensor(labels, dtype=torch.float64))
        loss.backward()
        opt.step()
        assert abs(loss.item() - ref_loss) < 1e-12
    for got, ref in ((st.PM, tp), (st.RE, tr), (st.CE, tc)):
        np.testing.assert_allclose(got, ref.detach().numpy(), rtol=1e-9, atol=1e-12)
