"""ctypes binding of ``libm2d.so`` (the C ABI declared in ``include/m2d.h``).

There is deliberately no fallback: if the HIP library is missing, or no MI355X is visible, every
entry point raises.  Nothing here (or anywhere in ``foodrec_amd``) imports ``oracle/``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libm2d.so")
CSRC = os.path.join(_PKG, "csrc")

M2D_OK = 0
M2D_ERR_INVALID_ARG = -1
M2D_ERR_HIP = -2
M2D_ERR_BAD_USER_ID = -3
M2D_ERR_BAD_ITEM_ID = -4
M2D_ERR_NOT_CONFIGURED = -5
M2D_ERR_UNSUPPORTED = -6
M2D_ERR_NO_DEVICE = -7
M2D_ERR_BAD_INGREDIENT = -8
M2D_ERR_KERNEL_TIMEOUT = -9          # m2d_topk_users: a wave gave up waiting for its workgroup; the lists are invalid
M2D_TABLES_HOST = 0
M2D_TABLES_DEVICE = 1
M2D_WRITE_PERSONAL = 1
M2D_WRITE_GENERAL = 2
ABI_VERSION = 2

_c = ctypes
_vp = _c.c_void_p
_i64 = _c.c_int64
_i32 = _c.c_int32

# name -> (restype, argtypes): exactly the declarations of include/m2d.h
SIGNATURES = {
    "m2d_abi_version": (_c.c_int, []),
    "m2d_create": (_c.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _c.c_float, _c.c_int, _c.c_int,
                              _c.POINTER(_vp)]),
    "m2d_destroy": (_c.c_int, [_vp]),
    "m2d_last_error": (_c.c_char_p, [_vp]),
    "m2d_last_kernel": (_c.c_char_p, [_vp]),
    "m2d_set_user_base": (_c.c_int, [_vp, _i64]),
    "m2d_set_dish_categories": (_c.c_int, [_vp, _vp, _c.c_int]),
    "m2d_score_pairs": (_c.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "m2d_score_pairs_host": (_c.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "m2d_score_pairs_bydish": (_c.c_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "m2d_rank_candidates": (_c.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "m2d_topk_users": (_c.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "m2d_set_ingredients": (_c.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _c.c_int]),
    "m2d_clear_ingredients": (_c.c_int, [_vp]),
    "m2d_score_pairs_ingredients": (_c.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "m2d_set_mlp_head": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _c.c_float, _i32, _i32, _c.c_int]),
    "m2d_clear_mlp_head": (_c.c_int, [_vp]),
    "m2d_score_pairs_mlp": (_c.c_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "m2d_write_memory": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _c.c_float, _c.c_float,
                                    _c.c_float, _i32, _vp, _vp]),
    "m2d_train_begin": (_c.c_int, [_vp, _i32, _c.c_float, _c.c_float, _vp]),
    "m2d_train_step": (_c.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "m2d_train_slot": (_c.c_int, [_vp, _i32, _i32, _vp, _i32, _vp]),
    "m2d_tables_updated": (_c.c_int, [_vp]),
    "m2d_train_steps": (_c.c_int, [_vp, _c.POINTER(_i64), _i32]),
    "m2d_train_end": (_c.c_int, [_vp]),
    "m2d_check": (_c.c_int, [_vp, _vp, _c.POINTER(_i64), _c.POINTER(_i64)]),
    "m2d_stream_read_probe": (_c.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "m2d_set_option": (_c.c_int, [_vp, _c.c_char_p, _i64]),
    "m2d_get_option": (_c.c_int, [_vp, _c.c_char_p, _c.POINTER(_i64)]),
}

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into foodrec_amd/libm2d.so (hipcc cross-compiles on CPU)."""
    cmd = ["make", "-C", CSRC, "-j", "4"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("hipcc build of libm2d.so failed")
    return LIB_PATH


def lib():
    """Load libm2d.so; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            "%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(foodrec_amd has no CPU fallback)" % LIB_PATH)
    # libm2d.so must share ONE HIP runtime with PyTorch (streams and events cross the boundary).
    # torch bundles its own libamdhip64.so.7; importing torch first makes the loader resolve
    # libm2d.so's NEEDED entry to that already-loaded copy instead of mapping /opt/rocm's as a second
    # runtime (which would see torch's stream handles as garbage).
    import torch  # noqa: F401
    l = ctypes.CDLL(LIB_PATH)
    with open("/proc/self/maps") as f:
        runtimes = {line.split()[-1] for line in f if "libamdhip64.so" in line}
    if len(runtimes) > 1:
        raise RuntimeError("two HIP runtimes are mapped in this process: %s" % sorted(runtimes))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = l
    return _lib


def error_text(handle) -> str:
    msg = lib().m2d_last_error(handle)
    return msg.decode("utf-8", "replace") if msg else ""


def raise_for(rc: int, handle=None):
    """Map an ABI status to the exception the reference's TF path would surface."""
    if rc == M2D_OK:
        return
    msg = error_text(handle) or ("m2d error %d" % rc)
    if rc in (M2D_ERR_BAD_USER_ID, M2D_ERR_BAD_ITEM_ID, M2D_ERR_BAD_INGREDIENT):
        raise IndexError(msg)          # TF-CPU GatherV2: InvalidArgumentError (indices out of range)
    if rc in (M2D_ERR_INVALID_ARG, M2D_ERR_UNSUPPORTED, M2D_ERR_NOT_CONFIGURED):
        raise ValueError(msg)
    raise RuntimeError(msg)
