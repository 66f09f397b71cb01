"""ctypes loader for oracle/flat_oracle.c.  TEST INFRASTRUCTURE ONLY (see the
header of flat_oracle.c): imported from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from the product package."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libflat_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "flat_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        f32p = ctypes.POINTER(ctypes.c_float)
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.oracle_knn_flat.restype = ctypes.c_int
        L.oracle_knn_flat.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, ctypes.c_int64,
                                      ctypes.c_int, ctypes.c_int, f32p, i64p, ctypes.c_int]
        L.oracle_renorm_L2.restype = None
        L.oracle_renorm_L2.argtypes = [ctypes.c_size_t, ctypes.c_size_t, f32p]
        L.oracle_fvec_L2sqr.restype = ctypes.c_float
        L.oracle_fvec_L2sqr.argtypes = [f32p, f32p, ctypes.c_size_t]
        L.oracle_fvec_inner_product.restype = ctypes.c_float
        L.oracle_fvec_inner_product.argtypes = [f32p, f32p, ctypes.c_size_t]
        L.oracle_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def knn_flat(xb, xq, k: int, metric: int = 1, nthreads: int = 0):
    """Faiss small-batch IndexFlat search restated in C (float32).
    Returns (D, I, threads_used)."""
    xb = np.ascontiguousarray(xb, dtype=np.float32)
    xq = np.ascontiguousarray(xq, dtype=np.float32)
    nq, d = xq.shape
    n = xb.shape[0]
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    nt = lib().oracle_knn_flat(_p(xb, ctypes.c_float), n, d, _p(xq, ctypes.c_float), nq, k, metric,
                               _p(D, ctypes.c_float), _p(I, ctypes.c_int64), nthreads)
    if nt < 0:
        raise MemoryError("oracle_knn_flat")
    return D, I, nt


def renorm_L2(x: np.ndarray) -> None:
    """In place, float32 C-contiguous (faiss.normalize_L2 contract)."""
    assert x.dtype == np.float32 and x.flags.c_contiguous and x.ndim == 2
    lib().oracle_renorm_L2(x.shape[1], x.shape[0], _p(x, ctypes.c_float))


def max_threads() -> int:
    return int(lib().oracle_max_threads())
