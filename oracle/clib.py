"""ctypes loader for oracle/csrc/oracle.cpp (test infrastructure)."""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "csrc", "oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "_build/liboracle.so"])
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_i64 = ctypes.c_int64


def sqnorm(x):
    x = _f(x)
    out = np.empty(x.shape[0], np.float32)
    lib().orc_sqnorm(_p(x), _i64(x.shape[0]), _i64(x.shape[1]), _p(out))
    return out


def l2_matrix(q, db):
    q, db = _f(q), _f(db)
    out = np.empty((q.shape[0], db.shape[0]), np.float32)
    lib().orc_l2_matrix(_p(q), _i64(q.shape[0]), _p(db), _i64(db.shape[0]), _i64(q.shape[1]), _p(out))
    return out


def l2_knn(q, db, k):
    q, db = _f(q), _f(db)
    d2 = np.empty((q.shape[0], k), np.float32)
    idx = np.empty((q.shape[0], k), np.int64)
    lib().orc_l2_knn(_p(q), _i64(q.shape[0]), _p(db), _i64(db.shape[0]), _i64(q.shape[1]), _i64(k), _p(d2), _p(idx))
    return d2, idx


def l2_argmin_cols(q, db):
    q, db = _f(q), _f(db)
    idx = np.empty(db.shape[0], np.int64)
    d2 = np.empty(db.shape[0], np.float32)
    lib().orc_l2_argmin_cols(_p(q), _i64(q.shape[0]), _p(db), _i64(db.shape[0]), _i64(q.shape[1]), _p(idx), _p(d2))
    return idx, d2


def dot_rows(bank, q, perm16=False):
    bank, q = _f(bank), _f(q)
    out = np.empty(bank.shape[0], np.float32)
    if perm16:
        slices = 8 if bank.shape[1] % 128 == 0 else 1
        lib().orc_dot_rows_perm16(_p(bank), _i64(bank.shape[0]), _i64(bank.shape[1]), _p(q), _i64(slices), _p(out))
    else:
        lib().orc_dot_rows(_p(bank), _i64(bank.shape[0]), _i64(bank.shape[1]), _p(q), _p(out))
    return out


def scatter_add(idx, src, size):
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    src = _f(src)
    out = np.zeros(size, np.float32)
    lib().orc_scatter_add(_p(idx), _p(src), _i64(idx.shape[0]), _p(out))
    return out


def topk_torch(values, k, largest=True, sorted_=True):
    values = _f(values)
    if not 0 <= k <= values.shape[0]:
        raise ValueError(f"k={k} out of range for {values.shape[0]} values (torch.topk raises too)")
    val = np.empty(k, np.float32)
    idx = np.empty(k, np.int64)
    lib().orc_topk_torch(_p(values), _i64(values.shape[0]), _i64(k), int(largest), int(sorted_), _p(val), _p(idx))
    return val, idx


def topk_canonical(values, k, largest=True):
    values = _f(values)
    if not 0 <= k <= values.shape[0]:
        raise ValueError(f"k={k} out of range for {values.shape[0]} values")
    val = np.empty(k, np.float32)
    idx = np.empty(k, np.int64)
    lib().orc_topk_canonical(_p(values), _i64(values.shape[0]), _i64(k), int(largest), _p(val), _p(idx))
    return val, idx
