"""ORACLE -- test infrastructure only; never imported by the product package.

ctypes faces of
  * oracle/_build/libknn_oracle.so  -- our C restatement (oracle/knn_oracle.c)
  * oracle/_ref/libknn_ref.so       -- the real reference nanoflann kNN
    (/root/reference/models/RandLA/utils/nearest_neighbors/knn_.cxx:104-135),
    present only where `make -C oracle ref` has been run (this container; the
    built .so travels to the GPU box with the gpurun snapshot).

Python face mirrors knn.pyx:71-109 `knn_batch(pts, queries, K, omp)` -> int64[B,Q,K]
and helper_tool.py:160-170 `DataProcessing.knn_search` (cast to int32).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "_build", "libknn_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libknn_ref.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(ref=True):
    """Compile the C restatement (and the reference, when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])
    if ref and os.path.isdir("/root/reference/models/RandLA/utils/nearest_neighbors"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_oracle = None
_ref = None


def _load_oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(_ORACLE_SO):
            build(ref=False)
        lib = ctypes.CDLL(_ORACLE_SO)
        lib.oracle_knn_batch.argtypes = [_f32p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                         _f32p, ctypes.c_size_t, ctypes.c_size_t, _i64p, _f32p]
        lib.oracle_knn_batch.restype = None
        lib.oracle_knn_d2_of.argtypes = [_f32p, _f32p, ctypes.c_size_t, ctypes.c_size_t, _i64p, _f32p]
        lib.oracle_knn_d2_of.restype = None
        _oracle = lib
    return _oracle


def have_ref():
    return os.path.exists(_REF_SO)


def _load_ref():
    global _ref
    if _ref is None:
        lib = ctypes.CDLL(_REF_SO)
        for name in ("ref_knn_batch_omp", "ref_knn_batch"):
            fn = getattr(lib, name)
            fn.argtypes = [_f32p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                           _f32p, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_long)]
            fn.restype = None
        _ref = lib
    return _ref


def _prep(pts, queries):
    pts = np.ascontiguousarray(pts, dtype=np.float32)       # knn.pyx:95
    queries = np.ascontiguousarray(queries, dtype=np.float32)  # knn.pyx:96
    assert pts.ndim == 3 and queries.ndim == 3 and pts.shape[2] == 3 and queries.shape[2] == 3
    assert pts.shape[0] == queries.shape[0]
    return pts, queries


def knn_batch(pts, queries, K, return_d2=False):
    """Oracle restatement. pts f32[B,S,3], queries f32[B,Q,3] -> int64[B,Q,K] (+ f32 d2)."""
    pts, queries = _prep(pts, queries)
    B, S, _ = pts.shape
    Q = queries.shape[1]
    idx = np.zeros((B, Q, K), dtype=np.int64)               # knn.pyx:93 zero-init
    d2 = np.zeros((B, Q, K), dtype=np.float32)
    _load_oracle().oracle_knn_batch(pts.ctypes.data_as(_f32p), B, S, 3,
                                    queries.ctypes.data_as(_f32p), Q, K,
                                    idx.ctypes.data_as(_i64p), d2.ctypes.data_as(_f32p))
    return (idx, d2) if return_d2 else idx


def ref_knn_batch(pts, queries, K, omp=True):
    """The REAL reference (nanoflann) through oracle/_ref. Same contract as knn.pyx:71-109."""
    pts, queries = _prep(pts, queries)
    B, S, _ = pts.shape
    Q = queries.shape[1]
    idx = np.zeros((B, Q, K), dtype=np.int64)
    fn = _load_ref().ref_knn_batch_omp if omp else _load_ref().ref_knn_batch
    fn(pts.ctypes.data_as(_f32p), B, S, 3, queries.ctypes.data_as(_f32p), Q, K,
       idx.ctypes.data_as(ctypes.POINTER(ctypes.c_long)))
    return idx


def d2_of(sup, queries, idx):
    """fp32 squared distances (reference arithmetic) of given neighbour indices. sup [S,3], queries [Q,3], idx [Q,K]."""
    sup = np.ascontiguousarray(sup, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    Q, K = idx.shape
    out = np.zeros((Q, K), dtype=np.float32)
    _load_oracle().oracle_knn_d2_of(sup.ctypes.data_as(_f32p), queries.ctypes.data_as(_f32p), Q, K,
                                    idx.ctypes.data_as(_i64p), out.ctypes.data_as(_f32p))
    return out


def knn_search(support_pts, query_pts, k):
    """helper_tool.py:160-170: int32 cast of knn_batch."""
    return knn_batch(support_pts, query_pts, k).astype(np.int32)
