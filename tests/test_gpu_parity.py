"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.

Tolerance (north_star: "within 1e-4 fp32"): |got - ref| <= 1e-4 * max(1, |ref|) against the float64
restatement; NaN (0/0 on an empty mask, Model_Recommender.py:79/:92) must appear at the same pairs.
"""
import ctypes

import numpy as np
import pytest

from helpers import COEFS, assert_scores_close, assert_scores_match_nonfinite, random_case, score_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch


def _engine(PM, RE, CE, **kw):
    from foodrec_amd import ScoringEngine
    return ScoringEngine(PM, RE, CE, **kw)


def _run(eng, torch, users, items, cats):
    dev = eng.device
    out = eng.score_pairs(torch.as_tensor(users, dtype=torch.int32, device=dev),
                          torch.as_tensor(items, dtype=torch.int32, device=dev),
                          torch.as_tensor(cats, dtype=torch.float32, device=dev))
    eng.check()
    return out.cpu().numpy()


def test_native_library_is_what_runs(torch_cuda):
    import foodrec_amd
    maps = open("/proc/self/maps").read()
    assert "libm2d.so" in maps



@pytest.mark.parametrize("path", score_cases(), ids=lambda p: p.split("score_")[-1][:-4])
def test_golden_vectors(torch_cuda, path):
    from oracle import m2d_oracle as oracle
    z = np.load(path)
    coef = float(z["coef"])                                # high_level_score_coefficient of the case (Train_recommender.py:61-62)
    eng = _engine(z["PM"], z["RE"], z["CE"], coef=coef)
    got = _run(eng, torch_cuda, z["users"], z["items"], z["cats"])
    assert_scores_close(got, z["score_f64"], what="HIP vs frozen f64")
    assert_scores_close(got, oracle.inference_f64(z["PM"], z["RE"], z["CE"], z["users"], z["items"], z["cats"], coef),
                        what="HIP vs live oracle")
    if "hand" in z.files:
        assert abs(float(got[0]) - 3.4625) < 1e-6


@pytest.mark.parametrize("shape", [(11, 13, 4, 6), (257, 129, 4, 32), (1000, 500, 4, 64), (1000, 500, 4, 128),
                                   (300, 100, 4, 200), (64, 64, 4, 256), (50, 40, 4, 260), (30, 20, 4, 7),
                                   (40, 30, 3, 16), (40, 30, 9, 64), (25, 12, 1, 4)])
def test_seeded_shapes(torch_cuda, shape):
    """Every pair kernel (C = 4 throughput / latency forms, C != 4, odd embedding sizes) at every blend coefficient:
    `coef * high + (1 - coef) * low` with `1 - coef` taken in float32 (Model_Recommender.py:17, :95-96) -- the reference's
    flag default is 0.99 (Train_recommender.py:61-62); 0 and 1 switch a level off, 1.25 makes the low level's weight negative.
    One test per shape: the batch sizes (1, around a wavefront, 4097) and the six coefficients are looped inside (30 cases)."""
    from oracle import m2d_oracle as oracle
    U, I, C, E = shape
    for B in (1, 63, 64, 65, 4097):
        PM, RE, CE, users, items, cats = random_case(U, I, C, E, B, seed=U + E + B)
        if B > 3:
            cats[1] = 1.0
            cats[2] = 0.0
            cats[3] = np.linspace(0.25, 1.75, C)
        for coef in [0.99] + COEFS:
            eng = _engine(PM, RE, CE, coef=coef)
            got = _run(eng, torch_cuda, users, items, cats)
            assert_scores_close(got, oracle.inference_f64(PM, RE, CE, users, items, cats, coef),
                                what="%s B %d coef %g" % (shape, B, coef))
            eng.close()


@pytest.mark.parametrize("coef", [0.99, 0.0, 1.0, 1.25])
def test_kernel_variants_agree(torch_cuda, coef):
    from oracle import m2d_oracle as oracle
    for E in (32, 64, 128, 200):
        PM, RE, CE, users, items, cats = random_case(500, 300, 4, E, 20000, seed=E)     # > 8192: the throughput form
        ref = oracle.inference_f64(PM, RE, CE, users, items, cats, coef)
        eng = _engine(PM, RE, CE, coef=coef)
        outs = []
        eng.set_option("variant", 12)                      # the latency form (default up to 8192 pairs), forced
        outs.append(_run(eng, torch_cuda, users, items, cats))
        assert eng.last_kernel() == "m2d_score_pairs_c4_small"
        assert_scores_close(outs[0], ref, what="latency form E%d" % E)
        eng.set_option("variant", 0)
        assert np.array_equal(_run(eng, torch_cuda, users[:100], items[:100], cats[:100]), outs[0][:100], equal_nan=True)
        assert eng.last_kernel() == "m2d_score_pairs_c4_small"
        eng.set_option("variant", 11)                      # small batch through the throughput form
        assert np.array_equal(_run(eng, torch_cuda, users[:100], items[:100], cats[:100]), outs[0][:100], equal_nan=True)
        assert eng.last_kernel() == "m2d_score_pairs_c4"
        eng.set_option("variant", 0)
        for pf in (1, 2, 4):
            for nt in (0, 1):
                for bpc in (1, 8):
                    eng.set_option("prefetch", pf); eng.set_option("nt_loads", nt); eng.set_option("blocks_per_cu", bpc)
                    got = _run(eng, torch_cuda, users, items, cats)
                    assert_scores_close(got, ref, what="E%d pf%d nt%d" % (E, pf, nt))
                    outs.append(got)
        for o in outs[1:]:
            assert np.array_equal(o, outs[0], equal_nan=True), "variants must be bit-identical"
        eng.set_option("variant", 9)                       # force the generic kernel
        assert_scores_close(_run(eng, torch_cuda, users, items, cats), ref, what="generic E%d" % E)
        assert eng.last_kernel() == "m2d_score_pairs_generic"


def test_bydish_equals_explicit_feed(torch_cuda):
    torch = torch_cuda
    PM, RE, CE, users, items, _ = random_case(300, 200, 4, 64, 3000, seed=77)
    dish_cats = np.random.default_rng(3).integers(0, 2, (200, 4)).astype(np.float32)
    eng = _engine(PM, RE, CE)
    with pytest.raises(ValueError):
        eng.score_pairs_bydish(torch.as_tensor(users, device=eng.device), torch.as_tensor(items, device=eng.device))
    eng.set_dish_categories(dish_cats)
    a = eng.score_pairs_bydish(torch.as_tensor(users, device=eng.device), torch.as_tensor(items, device=eng.device))
    b = _run(eng, torch, users, items, dish_cats[items])
    eng.check()
    assert np.array_equal(a.cpu().numpy(), b, equal_nan=True)


def test_out_of_range_ids_raise_and_engine_recovers(torch_cuda):
    from oracle import m2d_oracle as oracle
    PM, RE, CE, users, items, cats = random_case(20, 10, 4, 64, 200, seed=5, zero_rows=False)
    eng = _engine(PM, RE, CE)
    bad_u = users.copy(); bad_u[17] = 20
    with pytest.raises(IndexError, match="user id 20 at position 17"):
        _run(eng, torch_cuda, bad_u, items, cats)
    bad_i = items.copy(); bad_i[150] = -1
    with pytest.raises(IndexError, match="item id -1 at position 150"):
        _run(eng, torch_cuda, users, bad_i, cats)
    assert_scores_close(_run(eng, torch_cuda, users, items, cats), oracle.inference_f64(PM, RE, CE, users, items, cats))


def test_empty_batch_and_user_shard_offset(torch_cuda):
    from oracle import m2d_oracle as oracle
    PM, RE, CE, users, items, cats = random_case(64, 32, 4, 64, 500, seed=8, zero_rows=False)
    eng = _engine(PM, RE, CE)
    assert _run(eng, torch_cuda, users[:0], items[:0], cats[:0]).shape == (0,)
    # a shard holding global users [1000, 1064): global ids in, same scores out
    shard = _engine(PM, RE, CE, user_base=1000)
    got = _run(shard, torch_cuda, users + 1000, items, cats)
    assert_scores_close(got, oracle.inference_f64(PM, RE, CE, users, items, cats))
    with pytest.raises(IndexError):
        _run(shard, torch_cuda, users, items, cats)          # local ids are out of this shard's range


def test_raw_c_abi_with_host_tables(torch_cuda):
    """Straight through ctypes: host tables copied by the engine (M2D_TABLES_HOST), device id buffers."""
    torch = torch_cuda
    from foodrec_amd import _native
    from oracle import m2d_oracle as oracle
    lib = _native.lib()
    PM, RE, CE, users, items, cats = random_case(100, 50, 4, 64, 1000, seed=12)
    h = ctypes.c_void_p()
    rc = lib.m2d_create(PM.ctypes.data, RE.ctypes.data, CE.ctypes.data, 100, 50, 4, 64, 0.99, 0,
                        _native.M2D_TABLES_HOST, ctypes.byref(h))
    assert rc == 0, _native.error_text(None)
    u = torch.as_tensor(users, device="cuda"); d = torch.as_tensor(items, device="cuda")
    c = torch.as_tensor(cats, device="cuda"); out = torch.empty(1000, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.m2d_score_pairs(h, u.data_ptr(), d.data_ptr(), c.data_ptr(), 1000, out.data_ptr(), st) == 0
    assert lib.m2d_check(h, st, None, None) == 0
    assert_scores_close(out.cpu().numpy(), oracle.inference_f64(PM, RE, CE, users, items, cats))
    assert lib.m2d_score_pairs(h, None, d.data_ptr(), c.data_ptr(), 10, out.data_ptr(), st) == _native.M2D_ERR_INVALID_ARG
    assert lib.m2d_score_pairs_bydish(h, u.data_ptr(), d.data_ptr(), 10, out.data_ptr(), st) == _native.M2D_ERR_NOT_CONFIGURED
    assert lib.m2d_destroy(h) == 0


def test_model_predict_surface(torch_cuda):
    """Model(args, ...) + Session.run([model.logits], feed_dict) exactly as evaluate.py:55-59 feeds it."""
    import types
    from foodrec_amd import Model, Session
    from oracle import m2d_oracle as oracle
    PM, RE, CE, users, items, cats = random_case(64, 48, 4, 200, 51, seed=21, zero_rows=False)
    args = types.SimpleNamespace(learner="adam", num_categories=4, num_users=64, num_labels=95, embed_size=200,
                                 lr=0.001, decay_steps=1000, decay_rate=1.0, high_level_score_coefficient=0.99,
                                 beta_1=0.01, beta_2=0.01, alpha=0.01)
    model = Model(args, PM, RE, CE, np.zeros((95, 5, 200), np.float32))
    sess = Session(model)
    feed = {model.user_input: [str(u) for u in users], model.item_input: [int(i) for i in items],
            model.labels: [0] * 51, model.categories: cats[:, :, None].tolist(), model.dropout_keep_prob: 1.0,
            model.is_training_flag: False}
    pred = sess.run([model.logits], feed)[0]
    assert pred.dtype == np.float32 and pred.shape == (51,)
    assert_scores_close(pred, oracle.inference_f64(PM, RE, CE, users, items, cats))
    with pytest.raises(IndexError):
        model.predict([64], [0], cats[:1])
    with pytest.raises(NotImplementedError):
        sess.run([model.labels], feed)
    with pytest.raises(NotImplementedError):      # CPU tensors never fall back to an eager path
        torch_cuda.ops.m2d.score_pairs(model.engine.id, torch_cuda.zeros(1, dtype=torch_cuda.int32),
                                       torch_cuda.zeros(1, dtype=torch_cuda.int32), torch_cuda.ones(1, 4))


def test_full_size_properties(torch_cuda):
    """BASELINE config 2 sizes (1M users x 100k dishes, E = 64): the oracle cannot score 4M pairs in
    seconds, so check (i) a random sample of pairs against the oracle on the gathered rows,
    (ii) permutation invariance, (iii) linearity in Personal_Memory -- all size-independent."""
    torch = torch_cuda
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, C, E, B = 1_000_000, 100_000, 4, 64, 1 << 22
    g = torch.Generator(device="cuda"); g.manual_seed(20260102)
    s = 1.0 / np.sqrt(E)
    PM = torch.randn((U, C + 1, E), generator=g, device="cuda") * s
    RE = torch.randn((I, E), generator=g, device="cuda") * s
    CE = torch.randn((C, E), generator=g, device="cuda") * s
    users = torch.randint(0, U, (B,), generator=g, device="cuda", dtype=torch.int32)
    items = torch.randint(0, I, (B,), generator=g, device="cuda", dtype=torch.int32)
    pat = torch.randint(1, 16, (B,), generator=g, device="cuda", dtype=torch.int32)
    cats = ((pat[:, None] >> torch.arange(4, device="cuda", dtype=torch.int32)[None, :]) & 1).float()
    eng = ScoringEngine(PM, RE, CE)
    out = eng.score_pairs(users, items, cats); eng.check()
    assert torch.isfinite(out).all()
    # (i) sample
    idx = torch.randint(0, B, (4096,), generator=g, device="cuda")
    su, si = users[idx].long(), items[idx].long()
    ref = oracle.inference_f64(PM[su].cpu().numpy(), RE[si].cpu().numpy(), CE.cpu().numpy(), np.arange(4096),
                               np.arange(4096), cats[idx].cpu().numpy())
    assert_scores_close(out[idx].cpu().numpy(), ref, what="full-size sample")
    # (ii) permutation
    perm = torch.randperm(B, generator=g, device="cuda")
    out_p = eng.score_pairs(users[perm].contiguous(), items[perm].contiguous(), cats[perm].contiguous()); eng.check()
    assert torch.equal(out_p, out[perm])
    # (iii) linearity: score(2*PM) = 2*score(PM) exactly (power-of-two scaling commutes with rounding)
    eng2 = ScoringEngine(PM * 2, RE, CE)
    out2 = eng2.score_pairs(users, items, cats); eng2.check()
    assert torch.equal(out2, out * 2)


@pytest.mark.parametrize("zero_copy", [2, 1, 0])
@pytest.mark.parametrize("B", [1, 51, 5000, 65536, 65537, 262144, 262145, 700001])      # > 262144: chunks through two pinned blocks
def test_host_buffer_call_equals_the_device_op(torch_cuda, B, zero_copy):
    """m2d_score_pairs_host (what Model.predict uses for host feeds) against the torch custom op: same kernel, same bits,
    whether the kernel works on the pinned block itself (feeds of up to 65 536 pairs) or on a staged copy of it."""
    import torch
    from foodrec_amd import ScoringEngine
    PM, RE, CE, users, items, cats = random_case(400, 300, 4, 64, B, seed=B)
    eng = ScoringEngine(PM, RE, CE)
    eng.set_option("host_zero_copy", zero_copy)
    dev = lambda a: torch.as_tensor(a, device="cuda")
    ref = eng.score_pairs(dev(users), dev(items), dev(cats)).cpu().numpy(); eng.check()
    got = eng.score_pairs_host(users, items, cats)
    assert np.array_equal(got, ref, equal_nan=True)
    assert eng.score_pairs_host(users[:0], items[:0], cats[:0]).shape == (0,)
    bad = items.copy(); bad[B // 2] = 300
    with pytest.raises(IndexError, match="item id 300 at position %d" % (B // 2)):
        eng.score_pairs_host(users, bad, cats)
    assert np.array_equal(eng.score_pairs_host(users, items, cats), ref, equal_nan=True)      # latch cleared, engine usable
    with pytest.raises(ValueError):
        eng.score_pairs_host(users, items[:-1] if B > 1 else np.zeros(2, np.int32), cats)


@pytest.mark.parametrize("E,B", [(64, 5000), (200, 700), (24, 300)])       # c4 kernel (full / partial groups) and the small-batch form
def test_rows_of_weight_zero_categories_are_not_needed(E, B):
    """The Personal_Memory row of a category whose mask weight is 0 is multiplied by 0 (Model_Recommender.py:82): the
    kernels do not fetch it (option skip_masked, default 1) while every table value is finite.  Same scores either way
    for finite tables; with a non-finite value in a table the rows are fetched whatever the option says."""
    import torch
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, C = 400, 300, 4
    PM, RE, CE, users, items, cats = random_case(U, I, C, E, B, seed=E + 1)
    rng = np.random.default_rng(E)
    cats = (rng.integers(1, 16, B)[:, None] >> np.arange(C)[None, :] & 1).astype(np.float32)      # non-empty 0/1 masks
    cats[::7] *= rng.uniform(0.5, 2.0, (len(cats[::7]), C)).astype(np.float32)                    # some weighted masks
    eng = ScoringEngine(PM, RE, CE)
    ut, it, ct = (torch.as_tensor(x, device="cuda") for x in (users, items, cats))
    on = eng.score_pairs(ut, it, ct).cpu().numpy(); eng.check()
    eng.set_option("skip_masked", 0)
    off = eng.score_pairs(ut, it, ct).cpu().numpy(); eng.check()
    assert np.array_equal(on, off)                                          # the skipped terms are exact zeros
    assert_scores_close(on, oracle.inference_f64(PM, RE, CE, users, items, cats))
    # a non-finite value in the row of a category pair 0 does not have: 0 * inf = NaN in the graph (:82), and in the DEFAULT
    # configuration here -- the table scan finds it and the kernels fetch every row (tests/test_gpu_nonfinite.py)
    c0 = int(np.flatnonzero(cats[0] == 0)[0]) if (cats[0] == 0).any() else None
    if c0 is not None:
        PM2 = PM.copy(); PM2[users[0], c0 + 1, :] = np.inf
        eng2 = ScoringEngine(PM2, RE, CE)
        assert eng2.get_option("skip_masked") == 1
        got = eng2.score_pairs(ut[:1], it[:1], ct[:1]).cpu().numpy(); eng2.check()
        assert np.isnan(got[0]) and np.isnan(oracle.inference_f64(PM2, RE, CE, users[:1], items[:1], cats[:1])[0])
        assert_scores_match_nonfinite(eng2.score_pairs(ut, it, ct).cpu().numpy(), oracle.inference_f64(PM2, RE, CE, users, items, cats))


@pytest.mark.parametrize("coef", [0.99, 0.5, 1.25])
def test_user_high_table_option(torch_cuda, coef):
    """Serving option "user_high_table": batches of >= 2^18 pairs take the high-level sum from the derived table
    <U_high[u], CE_c>.  Same scores within rounding; the table follows the engine's tables (its own writers reset it,
    in-place edits from outside are announced with tables_updated() as for the retrieval tables)."""
    torch = torch_cuda
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, C, E, B = 700, 300, 4, 64, (1 << 18) + 77
    PM, RE, CE, users, items, cats = random_case(U, I, C, E, B, seed=21)
    pmt = torch.as_tensor(PM, device="cuda")
    eng = ScoringEngine(pmt, RE, CE, coef=coef)
    ut, it, ct = (torch.as_tensor(x, device="cuda") for x in (users, items, cats))
    lit = eng.score_pairs(ut, it, ct).cpu().numpy(); eng.check()
    assert eng.last_kernel() == "m2d_score_pairs_c4"
    eng.set_option("user_high_table", 1)
    tab = eng.score_pairs(ut, it, ct).cpu().numpy(); eng.check()
    assert eng.last_kernel() == "m2d_score_pairs_c4_uh"
    ok = ~np.isnan(lit)
    assert np.array_equal(np.isnan(tab), np.isnan(lit))
    assert np.max(np.abs(tab[ok] - lit[ok]) / np.maximum(1.0, np.abs(lit[ok]))) < 2e-6
    pick = np.arange(0, B, 97)
    assert_scores_close(tab[pick], oracle.inference_f64(PM, RE, CE, users[pick], items[pick], cats[pick], coef))
    eng.set_option("prefetch", 4)                                               # the table kernel's other instantiation
    tab4 = eng.score_pairs(ut, it, ct).cpu().numpy(); eng.check()
    assert eng.last_kernel() == "m2d_score_pairs_c4_uh" and np.array_equal(tab4, tab, equal_nan=True)
    eng.set_option("prefetch", 2)
    small = eng.score_pairs(ut[:5000], it[:5000], ct[:5000]).cpu().numpy()       # below the threshold: the literal kernels
    assert eng.last_kernel() == "m2d_score_pairs_c4_small" and np.array_equal(small, lit[:5000], equal_nan=True)
    # the tables change under the engine: announced -> the derived table is rebuilt
    pmt[:, 0, :] *= 2.0
    eng.tables_updated()
    PM2 = PM.copy(); PM2[:, 0, :] *= 2.0
    tab2 = eng.score_pairs(ut, it, ct).cpu().numpy(); eng.check()
    assert_scores_close(tab2[pick], oracle.inference_f64(PM2, RE, CE, users[pick], items[pick], cats[pick], coef))


def test_unusual_mask_weights_with_and_without_row_skipping(torch_cuda):
    """-0.0 counts as a zero weight (its products are zeros of either sign), NaN and inf weights keep their rows: the
    default and the literal fetch-and-multiply agree on every such mask, NaNs included."""
    torch = torch_cuda
    from foodrec_amd import ScoringEngine
    from oracle import m2d_oracle as oracle
    U, I, C, E, B = 50, 40, 4, 64, 4096 + 33
    PM, RE, CE, users, items, cats = random_case(U, I, C, E, B, seed=5)
    rng = np.random.default_rng(6)
    cats = rng.integers(0, 2, (B, C)).astype(np.float32)
    cats[cats.sum(1) == 0, 2] = 1
    cats[5::50, 0] = -0.0
    cats[7::50, 1] = np.nan
    cats[9::50, 3] = np.inf
    cats[11::50, 2] = -1.5
    eng = ScoringEngine(PM, RE, CE)
    ut, it, ct = (torch.as_tensor(x, device="cuda") for x in (users, items, cats))
    for n in (B, 300):                                       # throughput form and latency form
        on = eng.score_pairs(ut[:n], it[:n], ct[:n]).cpu().numpy(); eng.check()
        eng.set_option("skip_masked", 0)
        off = eng.score_pairs(ut[:n], it[:n], ct[:n]).cpu().numpy(); eng.check()
        eng.set_option("skip_masked", 1)
        assert np.array_equal(on, off, equal_nan=True)
        with np.errstate(invalid="ignore", divide="ignore"):
            ref = oracle.inference_f64(PM, RE, CE, users[:n], items[:n], cats[:n])
        assert np.array_equal(np.isnan(on), np.isnan(ref))
        ok = np.isfinite(ref)
        assert_scores_close(on[ok], ref[ok])


def test_two_engines_on_two_streams(torch_cuda):
    """include/m2d.h: distinct engines / streams are independent.  Two engines fed from two side streams, launches
    interleaved, against the same calls on the default stream; and one engine moved between streams with a synchronise."""
    torch = torch_cuda
    from foodrec_amd import ScoringEngine
    cases = [random_case(900, 700, 4, E, 60000, seed=E) for E in (64, 128)]
    engs, refs, feeds = [], [], []
    for PM, RE, CE, users, items, cats in cases:
        eng = ScoringEngine(PM, RE, CE)
        dc = np.random.default_rng(5).integers(0, 2, (700, 4)).astype(np.float32); dc[dc.sum(1) == 0, 0] = 1
        eng.set_dish_categories(dc)
        u, d, m = (torch.as_tensor(x, device=eng.device) for x in (users, items, cats))
        tk = torch.arange(0, 900, dtype=torch.int32, device=eng.device)
        refs.append((eng.score_pairs(u, d, m).clone(), eng.score_pairs_bydish(u, d).clone(), [t.clone() for t in eng.topk_users(tk, 10)]))
        eng.check()
        engs.append(eng); feeds.append((u, d, m, tk))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for rep in range(6):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                u, d, m, tk = feeds[i]
                outs[i].append((engs[i].score_pairs(u, d, m), engs[i].score_pairs_bydish(u, d), engs[i].topk_users(tk, 10)))
    for st in streams:
        st.synchronize()
    for i in (0, 1):
        engs[i].check()
        for a, b, (ts, ti) in outs[i]:
            assert torch.equal(a.nan_to_num(3.0), refs[i][0].nan_to_num(3.0))
            assert torch.equal(b.nan_to_num(3.0), refs[i][1].nan_to_num(3.0))
            assert torch.equal(ti, refs[i][2][1]) and torch.equal(ts.nan_to_num(3.0), refs[i][2][0].nan_to_num(3.0))
    # one engine, the other stream, after a synchronise; host feeds go through the stream that is current
    with torch.cuda.stream(streams[1]):
        u, d, m, tk = feeds[0]
        a = engs[0].score_pairs(u, d, m)
        host = engs[0].score_pairs_host(cases[0][3][:51], cases[0][4][:51], cases[0][5][:51])
    streams[1].synchronize()
    assert torch.equal(a.nan_to_num(3.0), refs[0][0].nan_to_num(3.0))
    assert np.array_equal(host, refs[0][0][:51].cpu().numpy(), equal_nan=True)


@pytest.mark.parametrize("C,E", [(1, 64), (2, 128), (3, 64), (5, 32), (6, 64), (7, 200), (8, 16), (3, 36), (6, 256), (5, 4)])
def test_vectorised_kernel_for_other_category_counts(torch_cuda, C, E):
    """C != 4 (up to 8), E % 4 == 0, more than 8192 pairs: m2d_score_pairs_cn, the C = 4 kernel's layout with the category
    loop unrolled to 8 -- against the restatement, against the one-wave-per-pair kernel, per-pair and resident masks, with
    and without row skipping and non-temporal loads; bad ids latch as everywhere."""
    torch = torch_cuda
    from oracle import m2d_oracle as oracle
    B = 20011
    coef = ([0.99] + COEFS)[(C + E // 4) % 6]             # each shape at one of the blend coefficients
    PM, RE, CE, users, items, cats = random_case(700, 300, C, E, B, seed=C * 100 + E)
    cats[7] = np.linspace(0.25, 1.75, C)
    cats[9] = 0.0                                         # 0/0 -> NaN
    ref = oracle.inference_f64(PM, RE, CE, users, items, cats, coef)
    eng = _engine(PM, RE, CE, coef=coef)
    outs = []
    for skip in (1, 0):
        for nt in (1, 0):
            eng.set_option("skip_masked", skip); eng.set_option("nt_loads", nt)
            got = _run(eng, torch, users, items, cats)
            assert eng.last_kernel() == "m2d_score_pairs_cn"
            assert_scores_close(got, ref, what="cn C%d E%d skip%d nt%d" % (C, E, skip, nt))
            outs.append(got)
    for o in outs[1:]:
        assert np.array_equal(o, outs[0], equal_nan=True)              # finite tables: the skipped rows only added zeros
    eng.set_option("variant", 9)
    gen = _run(eng, torch, users, items, cats)
    assert eng.last_kernel() == "m2d_score_pairs_generic"
    assert_scores_close(gen, ref, what="generic C%d E%d" % (C, E))
    eng.set_option("variant", 0)
    assert _run(eng, torch, users[:100], items[:100], cats[:100]).shape == (100,)
    assert eng.last_kernel() == "m2d_score_pairs_generic"             # small batches keep the one-wave-per-pair kernel
    dish_cats = np.random.default_rng(C).integers(0, 2, (300, C)).astype(np.float32)
    eng.set_dish_categories(dish_cats)
    u, d = torch.as_tensor(users, device=eng.device), torch.as_tensor(items, device=eng.device)
    byd = eng.score_pairs_bydish(u, d); eng.check()
    assert eng.last_kernel() == "m2d_score_pairs_cn"
    assert_scores_close(byd.cpu().numpy(), oracle.inference_f64(PM, RE, CE, users, items, dish_cats[items], coef), what="by dish")
    bad = items.copy(); bad[12345] = 300
    eng.score_pairs(u, torch.as_tensor(bad, device=eng.device), torch.as_tensor(cats, device=eng.device))
    with pytest.raises(IndexError, match="item id 300 at position 12345"):
        eng.check()
