"""TEST INFRASTRUCTURE ONLY -- ctypes loader for oracle/libsd_oracle.so (plain-C BCSD restatement)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libsd_oracle.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_PATH)
        _lib.sdo_bcsd_fit_predict.restype = C.c_int
        _lib.sdo_bcsd_fit_predict.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                                                             C.c_void_p, C.c_void_p, C.c_int]
        _lib.sdo_max_threads.restype = C.c_int
    return _lib


def max_threads():
    return load().sdo_max_threads()


def bcsd_fit_predict(kind, X, y, Xp, gid, gid_p, G=12, return_anoms=True, nthreads=1):
    lib = load()
    y = np.ascontiguousarray(y, dtype=np.float64)
    X = None if X is None else np.ascontiguousarray(X, dtype=np.float64)
    Xp = np.ascontiguousarray(Xp, dtype=np.float64)
    gid = np.ascontiguousarray(gid, dtype=np.int32)
    gid_p = np.ascontiguousarray(gid_p, dtype=np.int32)
    T, Cc = y.shape
    Tp = Xp.shape[0]
    out = np.empty((Tp, Cc))
    status = np.empty(Cc, dtype=np.int32)
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = lib.sdo_bcsd_fit_predict(kind, p(X), p(y), p(Xp), p(gid), p(gid_p), G, T, Tp, Cc, int(return_anoms), p(out), p(status), nthreads)
    assert rc == 0
    return out, status
