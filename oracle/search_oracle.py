"""oracle/search_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Two CPU restatements of memex's vector-search arithmetic:

* :func:`dist_cosine_np` / :func:`search_np` -- pure NumPy, element-order faithful, for small cases;
* :class:`COracle` -- ctypes binding of ``oracle/cosine_oracle.c`` (same arithmetic, OpenMP over
  queries) for sizes up to ~1e6 rows.

Reference semantics restated (reference = /root/reference, Rust, not buildable here):

* ids are dense and 1-based in insertion order   lib/libmemex/src/storage/local.rs:63
* search returns neighbours ascending by distance lib/libmemex/src/storage/local.rs:76-88
* ``similarity = 1.0 - (1.0 / (1.0 / distance))`` lib/libmemex/src/storage/local.rs:86
* distance = hnsw_rs 0.1.20 ``DistCosine`` (Cargo.lock:1716-1718; crate source not vendored):
  f32 products accumulated sequentially in f64, ``max(1 - dot/sqrt(na*nb), 0) as f32``,
  0 when either norm is 0.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def dist_cosine_np(a: np.ndarray, b: np.ndarray) -> np.float32:
    """hnsw_rs DistCosine::eval for f32 slices, element by element."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    assert a.shape == b.shape and a.ndim == 1
    dot = 0.0
    na = 0.0
    nb = 0.0
    for x, y in zip(a, b):
        dot += float(np.float32(x * y))
        na += float(np.float32(x * x))
        nb += float(np.float32(y * y))
    if na > 0.0 and nb > 0.0:
        d = 1.0 - dot / float(np.sqrt(np.float64(na * nb)))
        return np.float32(max(d, 0.0))
    return np.float32(0.0)


def score_from_dist(dist) -> np.ndarray:
    """local.rs:86 with every operation rounded to f32 (dist 0 -> 1/0=inf -> 1/inf=0 -> 1)."""
    d = np.asarray(dist, dtype=np.float32)
    with np.errstate(divide="ignore"):
        t = (np.float32(1.0) / d).astype(np.float32)
        u = (np.float32(1.0) / t).astype(np.float32)
    return (np.float32(1.0) - u).astype(np.float32)


def search_np(corpus: np.ndarray, queries: np.ndarray, k: int, id_offset: int = 0):
    """Exact brute force ordered by (dist_f32, id); pure Python loops -- small cases only."""
    corpus = np.asarray(corpus, dtype=np.float32)
    queries = np.asarray(queries, dtype=np.float32)
    n = corpus.shape[0]
    out_ids, out_d = [], []
    for q in queries:
        hits = sorted((float(dist_cosine_np(q, corpus[r])), r + 1 + id_offset) for r in range(n))[:k]
        out_ids.append([h[1] for h in hits] + [0] * (k - len(hits)))
        out_d.append([np.float32(h[0]) for h in hits] + [np.float32(np.inf)] * (k - len(hits)))
    ids = np.asarray(out_ids, dtype=np.uint64).reshape(len(queries), k)
    d = np.asarray(out_d, dtype=np.float32).reshape(len(queries), k)
    return ids, d, score_from_dist(d) * (ids != 0)


def build() -> str:
    """Compile oracle/cosine_oracle.c (gcc) -> oracle/libmxoracle.so.  Returns the path."""
    so = os.path.join(_HERE, "libmxoracle.so")
    src = os.path.join(_HERE, "cosine_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


class COracle:
    """ctypes view of oracle/libmxoracle.so."""

    def __init__(self) -> None:
        self.lib = ctypes.CDLL(build())
        L = self.lib
        fp = ctypes.POINTER(ctypes.c_float)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        L.mxo_dist_cosine.restype = ctypes.c_float
        L.mxo_dist_cosine.argtypes = [fp, fp, ctypes.c_int]
        L.mxo_score.restype = ctypes.c_float
        L.mxo_score.argtypes = [ctypes.c_float]
        L.mxo_search.restype = ctypes.c_int
        L.mxo_search.argtypes = [fp, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, fp, ctypes.c_int,
                                 ctypes.c_int, u64p, fp, fp, ctypes.POINTER(ctypes.c_int)]
        L.mxo_all_dists.restype = None
        L.mxo_all_dists.argtypes = [fp, ctypes.c_uint64, ctypes.c_int, fp, fp]
        L.mxo_merge.restype = ctypes.c_int
        L.mxo_merge.argtypes = [u64p, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, u64p, fp]
        L.mxo_num_threads.restype = ctypes.c_int

    @staticmethod
    def _f(a):
        return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))

    def num_threads(self) -> int:
        return int(self.lib.mxo_num_threads())

    def dist(self, a, b) -> np.float32:
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        return np.float32(self.lib.mxo_dist_cosine(self._f(a), self._f(b), a.shape[0]))

    def score(self, d) -> np.float32:
        return np.float32(self.lib.mxo_score(float(d)))

    def search(self, corpus, queries, k: int, id_offset: int = 0):
        corpus = np.ascontiguousarray(corpus, dtype=np.float32)
        queries = np.ascontiguousarray(queries, dtype=np.float32)
        if queries.ndim == 1:
            queries = queries[None, :]
        n, d = corpus.shape if corpus.ndim == 2 else (0, queries.shape[1])
        B = queries.shape[0]
        ids = np.zeros((B, k), dtype=np.uint64)
        dists = np.zeros((B, k), dtype=np.float32)
        scores = np.zeros((B, k), dtype=np.float32)
        nf = np.zeros(B, dtype=np.int32)
        rc = self.lib.mxo_search(self._f(corpus), n, d, id_offset, self._f(queries), B, k,
                                 ids.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), self._f(dists),
                                 self._f(scores), nf.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        if rc != 0:
            raise RuntimeError("mxo_search failed")
        return ids, dists, scores, nf

    def all_dists(self, corpus, q) -> np.ndarray:
        corpus = np.ascontiguousarray(corpus, dtype=np.float32)
        q = np.ascontiguousarray(q, dtype=np.float32)
        out = np.empty(corpus.shape[0], dtype=np.float32)
        self.lib.mxo_all_dists(self._f(corpus), corpus.shape[0], corpus.shape[1], self._f(q), self._f(out))
        return out

    def merge(self, ids, dists):
        """ids/dists: [G, B, k] per-shard lists -> merged [B, k] by (dist, id)."""
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        dists = np.ascontiguousarray(dists, dtype=np.float32)
        G, B, k = ids.shape
        oi = np.zeros((B, k), dtype=np.uint64)
        od = np.zeros((B, k), dtype=np.float32)
        self.lib.mxo_merge(ids.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), self._f(dists), G, B, k,
                           oi.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), self._f(od))
        return oi, od
