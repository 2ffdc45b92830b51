"""PyTorch-ROCm custom ops over the C ABI (``include/m2d.h``).

torch is plumbing here -- device memory, streams, ``torch.distributed`` -- the arithmetic is in
``libm2d.so``.  Ops are registered for the ``cuda`` (= HIP on ROCm) device only: handing them CPU
tensors raises ``NotImplementedError``; there is no eager/CPU fallback to fall through to.
"""
from __future__ import annotations

import ctypes
import itertools
import weakref
from typing import Optional

import numpy as np
import torch

from . import _native

_ENGINES: "weakref.WeakValueDictionary[int, ScoringEngine]" = weakref.WeakValueDictionary()
_ENGINE_SERIAL = itertools.count(1)         # engine keys are never reused (id() of a collected engine can be)


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _dev_f32(x, device) -> torch.Tensor:
    t = torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x)
    return t.to(device=device, dtype=torch.float32).contiguous()


class ScoringEngine:
    """Owns one ``m2d_engine`` and the HBM tables it borrows.

    Stands in for the variable set the reference builds in ``instantiate_weights``
    (Model_Recommender.py:43-54); ``user_base`` makes it one user-range shard.
    """

    def __init__(self, Personal_Memory, Recipe_Embedding, Category_Embedding, coef: float = 0.99,
                 device: Optional[torch.device] = None, user_base: int = 0):
        lib = _native.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("foodrec_amd needs an MI355X (no HIP device visible); there is no CPU fallback")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.pm = _dev_f32(Personal_Memory, self.device)
        self.re = _dev_f32(Recipe_Embedding, self.device)
        self.ce = _dev_f32(Category_Embedding, self.device)
        if self.pm.dim() != 3 or self.re.dim() != 2 or self.ce.dim() != 2:
            raise ValueError("Personal_Memory [U, C+1, E], Recipe_Embedding [I, E], Category_Embedding [C, E] expected")
        self.U, c1, self.E = self.pm.shape
        self.I = self.re.shape[0]
        self.C = self.ce.shape[0]
        if c1 != self.C + 1 or self.re.shape[1] != self.E or self.ce.shape[1] != self.E:
            raise ValueError("table shapes disagree: PM %s, RE %s, CE %s" % (tuple(self.pm.shape), tuple(self.re.shape), tuple(self.ce.shape)))
        self.coef = float(np.float32(coef))
        self.dish_cats: Optional[torch.Tensor] = None
        handle = ctypes.c_void_p()
        rc = lib.m2d_create(self.pm.data_ptr(), self.re.data_ptr(), self.ce.data_ptr(), self.U, self.I, self.C,
                            self.E, self.coef, self.device.index or 0, _native.M2D_TABLES_DEVICE,
                            ctypes.byref(handle))
        _native.raise_for(rc, None)
        self._h = handle
        self._host_fn = lib.m2d_score_pairs_host
        self.user_base = 0
        if user_base:
            self.set_user_base(user_base)
        self.id = next(_ENGINE_SERIAL)
        _ENGINES[self.id] = self

    # -- configuration ---------------------------------------------------------------------------
    def set_user_base(self, user_base: int):
        _native.raise_for(_native.lib().m2d_set_user_base(self._h, int(user_base)), self._h)
        self.user_base = int(user_base)

    def set_dish_categories(self, cats):
        """cats: [I, C] (or [I, C, 1]) float mask per dish -- dish_to_category.json as a table."""
        t = _dev_f32(cats, self.device).reshape(self.I, self.C)
        _native.raise_for(_native.lib().m2d_set_dish_categories(self._h, t.data_ptr(), _native.M2D_TABLES_DEVICE), self._h)
        self.dish_cats = t

    def set_ingredients(self, ing_table, offsets, ids, weights=None):
        """Build-defined extension: multi-hot ingredient lists per dish in CSR form.
        ing_table [R, E]; offsets i32[I+1]; ids i32[nnz]; weights f32[nnz] or None (all ones)."""
        t = _dev_f32(ing_table, self.device)
        if t.dim() != 2 or t.shape[1] != self.E:
            raise ValueError("ingredient table must be [R, E=%d]" % self.E)
        as_i32 = lambda x: torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x).to(
            device=self.device, dtype=torch.int32).contiguous()
        off, idt = as_i32(offsets), as_i32(ids)
        if off.numel() != self.I + 1:
            raise ValueError("offsets must have I + 1 = %d entries" % (self.I + 1))
        wt = _dev_f32(weights, self.device) if weights is not None else None
        if wt is not None and wt.numel() != idt.numel():
            raise ValueError("weights and ids differ in length")
        with torch.cuda.device(self.device):
            rc = _native.lib().m2d_set_ingredients(self._h, t.data_ptr(), t.shape[0], off.data_ptr(),
                                                   idt.data_ptr() if idt.numel() else None,
                                                   wt.data_ptr() if wt is not None else None, idt.numel(),
                                                   _native.M2D_TABLES_DEVICE)
        _native.raise_for(rc, self._h)
        self._ingredients = (t, off, idt, wt)          # keep the borrowed device buffers alive

    def clear_ingredients(self):
        _native.raise_for(_native.lib().m2d_clear_ingredients(self._h), self._h)
        self._ingredients = None

    def score_pairs_ingredients(self, users: torch.Tensor, items: torch.Tensor, cats: Optional[torch.Tensor] = None,
                                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Extension: high-level path from the ingredient table; cats=None -> resident dish masks."""
        self._check_ids(users, items)
        B = users.numel()
        users, items = users.contiguous(), items.contiguous()
        if cats is not None:
            if cats.numel() != B * self.C:
                raise ValueError("cats must be [B, C]")
            cats = cats.to(torch.float32).contiguous()
            if cats.data_ptr() % 16:
                cats = cats.clone()
        if out is None:
            out = torch.empty(B, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = _native.lib().m2d_score_pairs_ingredients(self._h, users.data_ptr(), items.data_ptr(),
                                                           cats.data_ptr() if cats is not None else None, B,
                                                           out.data_ptr(), _stream_ptr())
        _native.raise_for(rc, self._h)
        return out

    def set_mlp_head(self, W1, b1, W2, b2, w3, b3: float):
        """Build-defined extension: score = reference + w3.relu(W2^T relu(W1^T z + b1) + b2) + b3."""
        K = (self.C + 1) * self.E
        t = [_dev_f32(x, self.device) for x in (W1, b1, W2, b2, w3)]
        H1, H2 = t[0].shape[1], t[2].shape[1]
        if tuple(t[0].shape) != (K, H1) or tuple(t[2].shape) != (H1, H2) or t[1].numel() != H1 or t[3].numel() != H2 or t[4].numel() != H2:
            raise ValueError("MLP head shapes: W1 [K=%d, H1], b1 [H1], W2 [H1, H2], b2 [H2], w3 [H2]" % K)
        rc = _native.lib().m2d_set_mlp_head(self._h, *(x.data_ptr() for x in t), float(b3), H1, H2, _native.M2D_TABLES_DEVICE)
        _native.raise_for(rc, self._h)
        self._mlp = t

    def clear_mlp_head(self):
        _native.raise_for(_native.lib().m2d_clear_mlp_head(self._h), self._h)
        self._mlp = None

    def score_pairs_mlp(self, users: torch.Tensor, items: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Extension: reference score + MLP head, masks from the resident dish table."""
        self._check_ids(users, items)
        B = users.numel()
        users, items = users.contiguous(), items.contiguous()
        if out is None:
            out = torch.empty(B, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = _native.lib().m2d_score_pairs_mlp(self._h, users.data_ptr(), items.data_ptr(), B, out.data_ptr(), _stream_ptr())
        _native.raise_for(rc, self._h)
        return out

    def write_memory(self, users: torch.Tensor, items: torch.Tensor, cats: torch.Tensor, write_sign: torch.Tensor,
                     labels: torch.Tensor, general_memory: torch.Tensor, beta_1: float, beta_2: float, alpha: float,
                     want_means: bool = False, write_pm: bool = True, write_gm: bool = True):
        """Model.Write_Memory (Model_Recommender.py:106-220) as a scatter-add: updates self.pm (`write_pm`: the two
        chained Personal_Memory assigns the `personal` fetch depends on, :167 / :198) and `general_memory`
        ([L, C+1, E], device float32; `write_gm`: the assign the `general` fetch depends on, :215) IN PLACE.
        Returns (mean(PM), mean(GM)) when asked; the mean of a table that was not written is None."""
        if not (write_pm or write_gm):
            raise ValueError("write_memory: nothing to write")
        self._check_ids(users, items)
        B = users.numel()
        L = labels.shape[-1]
        if general_memory.device != self.device or general_memory.dtype != torch.float32 or not general_memory.is_contiguous():
            raise ValueError("general_memory must be a contiguous float32 tensor on %s" % self.device)
        if tuple(general_memory.shape) != (L, self.C + 1, self.E):
            raise ValueError("general_memory must be [L=%d, C+1, E]" % L)
        f = lambda t, n: t.to(device=self.device, dtype=torch.float32).reshape(B, n).contiguous()
        cats, write_sign, labels = f(cats, self.C), f(write_sign, 1), f(labels, L)
        sums = torch.zeros(2, dtype=torch.float64, device=self.device) if want_means else None
        with torch.cuda.device(self.device):
            rc = _native.lib().m2d_write_memory(self._h, users.contiguous().data_ptr(), items.contiguous().data_ptr(),
                                                cats.data_ptr(), write_sign.data_ptr(), labels.data_ptr(), B, L,
                                                general_memory.data_ptr(), float(beta_1), float(beta_2), float(alpha),
                                                (_native.M2D_WRITE_PERSONAL if write_pm else 0) |
                                                (_native.M2D_WRITE_GENERAL if write_gm else 0),
                                                sums.data_ptr() if want_means else None, _stream_ptr())
        _native.raise_for(rc, self._h)
        if want_means:
            self.check()
            s = sums.cpu().numpy()
            return (float(s[0] / self.pm.numel()) if write_pm else None,
                    float(s[1] / general_memory.numel()) if write_gm else None)
        return None

    # -- training step ----------------------------------------------------------
    LEARNERS = {"sgd": 0, "adagrad": 1, "rmsprop": 2, "adam": 3}

    def train_begin(self, learner: str = "adam", lr: float = 0.001, clip_norm: float = 5.0):
        """Model.train's optimizer choice (Model_Recommender.py:228-235): anything but adagrad / rmsprop / adam is
        gradient descent.  Allocates the optimizer slots; the three tables are updated in place from now on."""
        code = self.LEARNERS.get(str(learner).lower(), 0)
        with torch.cuda.device(self.device):
            rc = _native.lib().m2d_train_begin(self._h, code, float(lr), float(clip_norm), _stream_ptr())
        _native.raise_for(rc, self._h)
        self._training = True

    def train_step(self, users: torch.Tensor, items: torch.Tensor, cats: torch.Tensor, labels: torch.Tensor,
                   apply: bool = True) -> torch.Tensor:
        """One sess.run([loss_value, learning_rate, train_op]) (Train_recommender.py:189-199).  Returns a device
        float32[4]: loss, global gradient norm, clip scale, learning rate.  apply=False: loss / norm only."""
        self._check_ids(users, items)
        B = users.numel()
        # a step is two small launches at the reference's batch sizes: the wrapper's own cost counts.  Tensors that are
        # already what the kernel reads (contiguous float32 / int32 on the device) are passed as they are.
        ok = lambda t, n: (t.dtype == torch.float32 and t.device == self.device and t.is_contiguous() and t.numel() == B * n)
        if not ok(cats, self.C):
            cats = cats.to(device=self.device, dtype=torch.float32).reshape(B, self.C).contiguous()
        if not ok(labels, 1):
            labels = labels.to(device=self.device, dtype=torch.float32).reshape(B, 1).contiguous()
        if not users.is_contiguous():
            users = users.contiguous()
        if not items.is_contiguous():
            items = items.contiguous()
        out = torch.empty(4, dtype=torch.float32, device=self.device)
        fn = _native.lib().m2d_train_step
        idx = self.device.index
        if _raw_stream is not None and _cur_device is not None and idx is not None and _cur_device() == idx:
            rc = fn(self._h, users.data_ptr(), items.data_ptr(), cats.data_ptr(), labels.data_ptr(), B, 1 if apply else 0,
                    out.data_ptr(), _raw_stream(idx))
        else:
            with torch.cuda.device(self.device):
                rc = fn(self._h, users.data_ptr(), items.data_ptr(), cats.data_ptr(), labels.data_ptr(), B, 1 if apply else 0,
                        out.data_ptr(), _stream_ptr())
        if rc:
            _native.raise_for(rc, self._h)
        return out

    def train_slot(self, table: int, slot: int, restore: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Optimizer slot `slot` of table 0 = PM, 1 = RE, 2 = CE (adam: m, v; adagrad: accumulator; rmsprop: rms,
        momentum) as a tensor shaped like the table; with `restore`, that tensor is written INTO the slot instead
        (checkpoint resume)."""
        ref = (self.pm, self.re, self.ce)[table]
        if restore is not None:
            buf = restore.to(device=self.device, dtype=torch.float32).contiguous()
            if buf.shape != ref.shape:
                raise ValueError("slot must be shaped like its table %s" % (tuple(ref.shape),))
        else:
            buf = torch.empty_like(ref)
        with torch.cuda.device(self.device):
            rc = _native.lib().m2d_train_slot(self._h, int(table), int(slot), buf.data_ptr(), 0 if restore is None else 1,
                                              _stream_ptr())
        _native.raise_for(rc, self._h)
        return buf

    def tables_updated(self):
        """Call after writing to self.pm / self.re / self.ce directly (they are borrowed by the engine)."""
        _native.raise_for(_native.lib().m2d_tables_updated(self._h), self._h)

    def train_steps(self, restore: Optional[int] = None) -> int:
        """Optimizer steps applied since train_begin; with `restore`, sets that count (checkpoint resume)."""
        v = ctypes.c_int64(0 if restore is None else int(restore))
        _native.raise_for(_native.lib().m2d_train_steps(self._h, ctypes.byref(v), 0 if restore is None else 1), self._h)
        return int(v.value)

    def train_end(self):
        _native.raise_for(_native.lib().m2d_train_end(self._h), self._h)
        self._training = False

    def set_option(self, name: str, value: int):
        _native.raise_for(_native.lib().m2d_set_option(self._h, name.encode(), int(value)), self._h)

    def get_option(self, name: str) -> int:
        v = ctypes.c_int64()
        _native.raise_for(_native.lib().m2d_get_option(self._h, name.encode(), ctypes.byref(v)), self._h)
        return int(v.value)

    def last_kernel(self) -> str:
        return (_native.lib().m2d_last_kernel(self._h) or b"").decode()

    # -- raw launches (device tensors in, device tensors out, no sync) -----------------------------
    def _check_ids(self, users: torch.Tensor, items: torch.Tensor):
        if users.dtype != torch.int32 or items.dtype != torch.int32:
            raise TypeError("ids must be int32 tensors (Model_Recommender.py:26-29)")
        if users.device != self.device or items.device != self.device:
            raise NotImplementedError("m2d ops run on %s only; got %s" % (self.device, users.device))

    def score_pairs(self, users: torch.Tensor, items: torch.Tensor, cats: torch.Tensor,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._check_ids(users, items)
        B = users.numel()
        if items.numel() != B or cats.numel() != B * self.C:
            raise ValueError("score_pairs: users[%d], items[%d], cats[%d] disagree (C=%d)" % (B, items.numel(), cats.numel(), self.C))
        users, items = users.contiguous(), items.contiguous()
        cats = cats.to(torch.float32).contiguous()
        if cats.data_ptr() % 16:
            cats = cats.clone()
        if out is None:
            out = torch.empty(B, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = _native.lib().m2d_score_pairs(self._h, users.data_ptr(), items.data_ptr(), cats.data_ptr(), B,
                                               out.data_ptr(), _stream_ptr())
        _native.raise_for(rc, self._h)
        return out

    def score_pairs_host(self, users: np.ndarray, items: np.ndarray, cats: np.ndarray) -> np.ndarray:
        """Host arrays in, host array out (int32 [B], int32 [B], float32 [B, C] -> float32 [B]): the latency path for
        reference-shaped calls of a few dozen pairs.  Synchronous; raises IndexError for an out-of-range id."""
        users = np.ascontiguousarray(users, dtype=np.int32)
        items = np.ascontiguousarray(items, dtype=np.int32)
        cats = np.ascontiguousarray(cats, dtype=np.float32)
        B = users.shape[0]
        if items.shape != (B,) or cats.shape != (B, self.C):
            raise ValueError("users [B], items [B], cats [B, %d] expected" % self.C)
        out = np.empty(B, dtype=np.float32)
        ptr = lambda a: a.__array_interface__["data"][0]
        idx = self.device.index
        if _raw_stream is not None and _cur_device is not None and idx is not None and _cur_device() == idx:
            # a 51-pair call is a few tens of microseconds: skip the device guard and the Stream object when nothing needs them
            rc = self._host_fn(self._h, ptr(users), ptr(items), ptr(cats), B, ptr(out), _raw_stream(idx))
        else:
            with torch.cuda.device(self.device):
                rc = self._host_fn(self._h, ptr(users), ptr(items), ptr(cats), B, ptr(out), _stream_ptr())
        if rc:
            _native.raise_for(rc, self._h)
        return out

    def score_pairs_bydish(self, users: torch.Tensor, items: torch.Tensor,
                           out: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._check_ids(users, items)
        B = users.numel()
        if items.numel() != B:
            raise ValueError("score_pairs_bydish: users and items differ in length")
        users, items = users.contiguous(), items.contiguous()
        if out is None:
            out = torch.empty(B, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = _native.lib().m2d_score_pairs_bydish(self._h, users.data_ptr(), items.data_ptr(), B, out.data_ptr(),
                                                      _stream_ptr())
        _native.raise_for(rc, self._h)
        return out

    def rank_candidates(self, users: torch.Tensor, items: torch.Tensor, k: int,
                        lens: Optional[torch.Tensor] = None):
        """users i32[nseg], items i32[nseg, L] -> (scores f32[nseg, k], items i32[nseg, k], flags i32[nseg])."""
        if items.dim() != 2 or users.numel() != items.shape[0]:
            raise ValueError("rank_candidates: items must be [nseg, L] with one user per row")
        self._check_ids(users, items)
        nseg, L = items.shape
        users, items = users.contiguous(), items.contiguous()
        if lens is not None:
            lens = lens.to(device=self.device, dtype=torch.int32).contiguous()
        out_s = torch.empty((nseg, k), dtype=torch.float32, device=self.device)
        out_i = torch.empty((nseg, k), dtype=torch.int32, device=self.device)
        out_f = torch.empty((nseg,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = _native.lib().m2d_rank_candidates(self._h, users.data_ptr(), items.data_ptr(),
                                                   lens.data_ptr() if lens is not None else None, nseg, L, k,
                                                   out_s.data_ptr(), out_i.data_ptr(), out_f.data_ptr(), _stream_ptr())
        _native.raise_for(rc, self._h)
        return out_s, out_i, out_f

    def topk_users(self, users: torch.Tensor, k: int):
        """users i32[nU] -> (scores f32[nU, k], dish ids i32[nU, k]) over the whole catalogue."""
        nU = users.numel()
        out_s = torch.empty((nU, k), dtype=torch.float32, device=self.device)
        out_i = torch.empty((nU, k), dtype=torch.int32, device=self.device)
        return self.topk_users_into(users, k, out_s, out_i)

    def topk_users_into(self, users: torch.Tensor, k: int, out_s: torch.Tensor, out_i: torch.Tensor):
        """topk_users into caller-owned contiguous buffers f32[nU, k] / i32[nU, k] (slices of a larger result)."""
        if users.dtype != torch.int32 or users.device != self.device:
            raise TypeError("topk_users: users must be an int32 tensor on %s" % self.device)
        users = users.contiguous()
        nU = users.numel()
        if (tuple(out_s.shape) != (nU, k) or tuple(out_i.shape) != (nU, k) or out_s.dtype != torch.float32 or
                out_i.dtype != torch.int32 or not out_s.is_contiguous() or not out_i.is_contiguous() or
                out_s.device != self.device or out_i.device != self.device):
            raise ValueError("topk_users_into: need contiguous f32[%d, %d] / i32[%d, %d] buffers on %s" % (nU, k, nU, k, self.device))
        with torch.cuda.device(self.device):
            rc = _native.lib().m2d_topk_users(self._h, users.data_ptr(), nU, k, out_s.data_ptr(), out_i.data_ptr(),
                                              _stream_ptr())
        _native.raise_for(rc, self._h)
        return out_s, out_i

    def stream_read_probe(self, buf: torch.Tensor, sink: torch.Tensor):
        """Calibration only: plain streaming read of `buf` (achievable HBM read rate of this box)."""
        nbytes = buf.numel() * buf.element_size()
        with torch.cuda.device(self.device):
            rc = _native.lib().m2d_stream_read_probe(self._h, buf.data_ptr(), nbytes - nbytes % 16, sink.data_ptr(),
                                                     _stream_ptr())
        _native.raise_for(rc, self._h)

    def check(self):
        """Synchronise the current stream; raise IndexError for an out-of-range id seen by a kernel."""
        bv, bi = ctypes.c_int64(), ctypes.c_int64()
        with torch.cuda.device(self.device):
            rc = _native.lib().m2d_check(self._h, _stream_ptr(), ctypes.byref(bv), ctypes.byref(bi))
        _native.raise_for(rc, self._h)

    def close(self):
        if getattr(self, "_h", None):
            _native.lib().m2d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- torch.library registration -----------------------------------------------------------------
# The engine travels through the dispatcher as an integer key into _ENGINES.

def _engine(eid: int) -> ScoringEngine:
    try:
        return _ENGINES[eid]
    except KeyError:
        raise RuntimeError("m2d: engine %d is gone" % eid) from None


@torch.library.custom_op("m2d::score_pairs", mutates_args=(), device_types="cuda")
def score_pairs_op(engine: int, users: torch.Tensor, items: torch.Tensor, cats: torch.Tensor) -> torch.Tensor:
    return _engine(engine).score_pairs(users, items, cats)


@score_pairs_op.register_fake
def _(engine, users, items, cats):
    return users.new_empty(users.shape, dtype=torch.float32)


@torch.library.custom_op("m2d::score_pairs_bydish", mutates_args=(), device_types="cuda")
def score_pairs_bydish_op(engine: int, users: torch.Tensor, items: torch.Tensor) -> torch.Tensor:
    return _engine(engine).score_pairs_bydish(users, items)


@score_pairs_bydish_op.register_fake
def _(engine, users, items):
    return users.new_empty(users.shape, dtype=torch.float32)


@torch.library.custom_op("m2d::topk_users", mutates_args=(), device_types="cuda")
def topk_users_op(engine: int, users: torch.Tensor, k: int) -> tuple[torch.Tensor, torch.Tensor]:
    return _engine(engine).topk_users(users, k)


@topk_users_op.register_fake
def _(engine, users, k):
    return (users.new_empty((users.numel(), k), dtype=torch.float32),
            users.new_empty((users.numel(), k), dtype=torch.int32))
