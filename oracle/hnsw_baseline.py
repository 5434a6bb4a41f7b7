"""oracle/hnsw_baseline.py -- TEST / BENCH INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes binding of ``oracle/hnsw_baseline.cpp``: the reference's real CPU search algorithm (hnsw_rs
HNSW with memex's parameters M=16, ef_construction=200, max_layer=16, search ef=32, DistCosine;
lib/libmemex/src/storage/local.rs:76,101) restated from the published algorithm.  Used by
``bench.py``'s ``cpu_baseline`` leg (QPS on one thread + recall@k against exact search) and by
``tests/test_hnsw_baseline.py``.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py`` may
import this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# memex's constants: local.rs:101 (M, max_layer, ef_construction) and local.rs:76 (ef)
M, MAX_LAYER, EF_CONSTRUCTION, EF_SEARCH = 16, 16, 200, 32


def _lib():
    path = os.path.join(_HERE, "libmxhnsw.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libmxhnsw.so"])
    L = ctypes.CDLL(path)
    L.mxh_build.restype = ctypes.c_void_p
    L.mxh_build.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                            ctypes.c_uint64, ctypes.c_int]
    L.mxh_search.restype = ctypes.c_double
    L.mxh_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                             ctypes.c_void_p]
    L.mxh_free.restype = None
    L.mxh_free.argtypes = [ctypes.c_void_p]
    return L


class HnswBaseline:
    def __init__(self, x: np.ndarray, seed: int = 1, threads: int = 0):
        """Build over ``x`` ([n, d] f32; kept referenced).  ``threads`` = build threads (0 = all)."""
        self._L = _lib()
        self.x = np.ascontiguousarray(x, dtype=np.float32)
        n, d = self.x.shape
        self._h = self._L.mxh_build(self.x.ctypes.data_as(ctypes.c_void_p), n, d, M, EF_CONSTRUCTION, MAX_LAYER, seed, threads)

    def search(self, q: np.ndarray, k: int, ef: int = EF_SEARCH):
        """-> (ids u64 [nq,k] 1-based, dists f32 [nq,k], seconds on ONE thread)."""
        q = np.ascontiguousarray(q, dtype=np.float32)
        nq = q.shape[0]
        ids = np.zeros((nq, k), dtype=np.uint64)
        dists = np.zeros((nq, k), dtype=np.float32)
        sec = self._L.mxh_search(self._h, q.ctypes.data_as(ctypes.c_void_p), nq, k, ef,
                                 ids.ctypes.data_as(ctypes.c_void_p), dists.ctypes.data_as(ctypes.c_void_p))
        return ids, dists, float(sec)

    def close(self):
        if self._h:
            self._L.mxh_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
