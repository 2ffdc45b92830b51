"""ctypes binding of oracle/libm2d_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY --
see m2d_oracle.c for the parity status (unpinned for the TF arithmetic)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libm2d_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """gcc the restatement next to its source (seconds)."""
    src = os.path.join(_HERE, "m2d_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libm2d_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        f32p = ctypes.POINTER(ctypes.c_float)
        i32p = ctypes.POINTER(ctypes.c_int32)
        common = [f32p, f32p, f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                  ctypes.c_float, i32p, i32p, f32p, ctypes.c_int64, f32p]
        _lib.m2d_oracle_score_pairs.argtypes = common + [ctypes.c_int]
        _lib.m2d_oracle_score_pairs.restype = ctypes.c_int
        _lib.m2d_oracle_score_pairs_materialised.argtypes = common + [f32p, ctypes.c_int]
        _lib.m2d_oracle_score_pairs_materialised.restype = ctypes.c_int
        _lib.m2d_oracle_max_threads.restype = ctypes.c_int
    return _lib


def max_threads() -> int:
    return int(lib().m2d_oracle_max_threads())


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def score_pairs(PM, RE, CE, users, items, cats, coef=0.99, nthreads=0, materialised=False):
    PM, RE, CE = _f32(PM), _f32(RE), _f32(CE)
    users = np.ascontiguousarray(users, dtype=np.int32)
    items = np.ascontiguousarray(items, dtype=np.int32)
    cats = _f32(cats).reshape(len(users), -1)
    U, C1, E = PM.shape
    C = CE.shape[0]
    assert C1 == C + 1 and cats.shape[1] == C and RE.shape[1] == E
    out = np.empty(len(users), dtype=np.float32)
    args = [_p(PM, ctypes.c_float), _p(RE, ctypes.c_float), _p(CE, ctypes.c_float), U, RE.shape[0], C, E,
            float(np.float32(coef)), _p(users, ctypes.c_int32), _p(items, ctypes.c_int32),
            _p(cats, ctypes.c_float), len(users), _p(out, ctypes.c_float)]
    if materialised:
        scratch = np.empty(4 * len(users) * C * E, dtype=np.float32)
        rc = lib().m2d_oracle_score_pairs_materialised(*args, _p(scratch, ctypes.c_float), nthreads)
    else:
        rc = lib().m2d_oracle_score_pairs(*args, nthreads)
    if rc == -1:
        raise IndexError("user id out of range")
    if rc == -2:
        raise IndexError("item id out of range")
    return out
