"""CPU checks of the boundary: libm2d.so loads, exports every symbol include/m2d.h declares, refuses
to run without a device (no fallback), and the host-side feed conversions behave like the int32 /
float32 feeds of the reference (Model_Recommender.py:26-32).  No compute call happens here."""
import ctypes
import os
import re
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "m2d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(m2d_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported(native_lib):
    from foodrec_amd import _native
    syms = declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(native_lib, s), "libm2d.so does not export %s" % s
    assert sorted(_native.SIGNATURES) == syms, "ctypes table and include/m2d.h disagree"
    assert native_lib.m2d_abi_version() == 2


def test_no_cpu_fallback(native_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from foodrec_amd import _native, ScoringEngine
    h = ctypes.c_void_p()
    buf = np.zeros(64, np.float32)
    rc = native_lib.m2d_create(buf.ctypes.data, buf.ctypes.data, buf.ctypes.data, 1, 1, 4, 4, 0.99, 0,
                               _native.M2D_TABLES_HOST, ctypes.byref(h))
    assert rc == _native.M2D_ERR_NO_DEVICE and not h.value
    assert b"no HIP device" in native_lib.m2d_last_error(None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ScoringEngine(np.zeros((2, 5, 8), np.float32), np.zeros((3, 8), np.float32), np.zeros((4, 8), np.float32))


def test_create_rejects_bad_arguments(native_lib):
    from foodrec_amd import _native
    h = ctypes.c_void_p()
    buf = np.zeros(64, np.float32)
    p = buf.ctypes.data
    assert native_lib.m2d_create(None, p, p, 1, 1, 4, 4, 0.99, 0, 0, ctypes.byref(h)) == _native.M2D_ERR_INVALID_ARG
    assert native_lib.m2d_create(p, p, p, 0, 1, 4, 4, 0.99, 0, 0, ctypes.byref(h)) == _native.M2D_ERR_INVALID_ARG
    assert native_lib.m2d_create(p, p, p, 1, 1, 4, 4, 0.99, 0, 7, ctypes.byref(h)) == _native.M2D_ERR_INVALID_ARG
    assert native_lib.m2d_create(p, p, p, 1, 2 ** 31, 4, 4, 0.99, 0, 0, ctypes.byref(h)) == _native.M2D_ERR_UNSUPPORTED
    assert native_lib.m2d_destroy(None) == 0


def test_feed_conversions():
    from foodrec_amd.recommender import _ids, _mask
    assert _ids(["3", "0", 7], "user").tolist() == [3, 0, 7] and _ids(["3"], "u").dtype == np.int32
    with pytest.raises(IndexError):
        _ids([2 ** 31], "user")
    nested = [[[1.0], [0.0], [1.0], [0.0]], [[0.5], [0.0], [0.0], [2.0]]]
    m = _mask(nested, 4, 2)
    assert m.shape == (2, 4) and m.dtype == np.float32 and m[1, 3] == 2.0
    with pytest.raises(ValueError):
        _mask(nested, 4, 3)
    with pytest.raises(ValueError):
        _mask(np.zeros((2, 5)), 4, 2)


def test_error_mapping():
    from foodrec_amd import _native
    with pytest.raises(IndexError):
        _native.raise_for(_native.M2D_ERR_BAD_USER_ID)
    with pytest.raises(ValueError):
        _native.raise_for(_native.M2D_ERR_NOT_CONFIGURED)
    with pytest.raises(RuntimeError):
        _native.raise_for(_native.M2D_ERR_HIP)
    _native.raise_for(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "foodrec_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "libm2d_oracle" not in text, f


def test_eval_plan_fingerprint_is_exact_for_replaced_lists_and_cheap():
    """The evaluator's plan cache (foodrec_amd/evaluator.py): the check that decides whether a cached plan may be reused
    sees every replaced list -- whichever user it belongs to -- and costs a few milliseconds at the reference's 64 657
    users (Train_recommender.py:51)."""
    import time
    from foodrec_amd import evaluator
    U = 64657
    rng = np.random.default_rng(0)
    neg = rng.integers(0, 4548, (U, 100))
    ratings = {str(u): [int(neg[u, 0])] for u in range(U)}
    negatives = {str(u): neg[u].tolist() for u in range(U)}
    d2c = {str(d): [[1.0], [0.0], [0.0], [1.0]] for d in range(4548)}
    held, stamp = evaluator._held(ratings, negatives, d2c), evaluator._stamp(ratings, negatives, d2c)
    dt = 1.0
    for _ in range(7):
        t0 = time.perf_counter()
        same = held == evaluator._held(ratings, negatives, d2c) and stamp == evaluator._stamp(ratings, negatives, d2c)
        dt = min(dt, time.perf_counter() - t0)
    assert same and dt < 0.010, dt
    for u in ("1", "777", "12345", str(U - 1)):              # users a 64-user sample would not look at
        old = negatives[u]
        negatives[u] = old[:50] + old[50:][::-1]
        assert held != evaluator._held(ratings, negatives, d2c)
        negatives[u] = list(old)                             # equal content in a new object: the same split
        assert held == evaluator._held(ratings, negatives, d2c)
    ratings["4242"] = [ratings["4242"][0] + 1]
    assert held != evaluator._held(ratings, negatives, d2c)
