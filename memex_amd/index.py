"""Thin object wrapper over the ``mx_index_*`` C ABI (include/memex_hip.h).

``FlatIndex`` is the GPU-resident exact cosine index that stands in for memex's ``HnswStore``
(reference lib/libmemex/src/storage/local.rs:21-166).  Host arrays are NumPy; the ``*_device``
methods take anything exposing ``data_ptr()`` (torch tensors already in HBM).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from ._lib import IndexStats, check, lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _caller_stream(t) -> ctypes.c_void_p | None:
    """The HIP stream torch is enqueueing work for tensor ``t`` on (None when torch is not in use)."""
    try:
        import torch
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    except ImportError:
        pass
    return None


SEARCH_AUTO, SEARCH_EXACT = 0, 1   # mx_index_set_search_mode


class FlatIndex:
    """``devices=None``: one index on ``device``.  ``devices=[...]``: the in-library sharded index
    (``mx_index_open_sharded``): rows dealt to the listed devices in blocks of ``block_rows``, local
    scans in parallel, one exchange of the per-shard top-k, merge on ``devices[0]``."""

    def __init__(self, dim: int, key: str | None = None, device: int = 0, devices=None, block_rows: int = 0):
        h = ctypes.c_void_p()
        if devices is None:
            check(lib().mx_index_open(key.encode() if key else None, int(dim), int(device), ctypes.byref(h)))
        else:
            devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
            check(lib().mx_index_open_sharded(key.encode() if key else None, int(dim), len(devices), devs,
                                              int(block_rows), ctypes.byref(h)))
            device = int(devices[0])
        self._h = h
        self.dim = int(dim)
        self.device = int(device)
        self.key = key

    @property
    def n_shards(self) -> int:
        n = ctypes.c_int(0)
        check(lib().mx_index_n_shards(self._h, ctypes.byref(n)))
        return int(n.value)

    @property
    def exchange(self) -> str:
        """How the shards exchange their top-k blocks: "none" (plain index), "p2p" (copies) or "rccl"."""
        kind = ctypes.c_int(0)
        check(lib().mx_index_exchange(self._h, ctypes.byref(kind)))
        return ("none", "p2p", "rccl")[int(kind.value)]

    def wait_stream(self, stream) -> None:
        """Order the next operation on this index after everything enqueued on ``stream`` (a raw
        hipStream_t as int / c_void_p, or None for the default stream).  See the stream contract in
        include/memex_hip.h; the ``*_device`` methods call this with torch's current stream."""
        check(lib().mx_index_wait_stream(self._h, stream))

    # -- lifetime -------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().mx_index_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __len__(self) -> int:
        n = ctypes.c_uint64(0)
        check(lib().mx_index_size(self._h, ctypes.byref(n)))
        return int(n.value)

    # -- configuration -------------------------------------------------------------------
    def reserve(self, n_rows: int) -> None:
        check(lib().mx_index_reserve(self._h, int(n_rows)))

    def set_id_offset(self, off: int) -> None:
        check(lib().mx_index_set_id_offset(self._h, int(off)))

    def set_search_mode(self, mode: int) -> None:
        check(lib().mx_index_set_search_mode(self._h, int(mode)))

    def set_filter_copy(self, on) -> None:
        """Keep or drop the filter copy the scan streams; results do not change.  ``False`` / ``"none"``: none (the
        scan reads the f32 rows); ``True`` / ``"auto"``: the library chooses (int8 up to 1024 dims, bf16 above, and an
        int8 copy is rebuilt as bf16 if a batch overflows it); ``"i8"``: int8 rows with one quantisation step per 32
        rows; ``"bf16"``: bf16 rows."""
        if isinstance(on, str):
            kind = {"none": 0, "auto": 1, "i8": 2, "int8": 2, "bf16": 3}[on]
        else:
            kind = 1 if on else 0
        check(lib().mx_index_set_filter_copy(self._h, int(kind)))

    def set_corpus_mode(self, mode: str) -> None:
        """``"f32"`` (default) or ``"bf16"``: keep only the bf16 rows (a third of the HBM); searches are
        then exact with respect to the stored rows (``get_rows``).  Only while the index is empty."""
        check(lib().mx_index_set_corpus_mode(self._h, {"f32": _lib.MX_CORPUS_F32, "bf16": _lib.MX_CORPUS_BF16}[mode]))

    def get_rows(self, first_row: int, n: int) -> np.ndarray:
        """Rows as stored (0-based, insertion order) -> f32 [n, dim]."""
        out = np.zeros((int(n), self.dim), dtype=np.float32)
        check(lib().mx_index_get_rows(self._h, int(first_row), int(n), _ptr(out)))
        return out

    def set_profiling(self, on: bool) -> None:
        check(lib().mx_index_set_profiling(self._h, 1 if on else 0))

    def stats(self) -> IndexStats:
        s = IndexStats()
        check(lib().mx_index_get_stats(self._h, ctypes.byref(s)))
        return s

    def reset_stats(self) -> None:
        check(lib().mx_index_reset_stats(self._h))

    # -- mutation ------------------------------------------------------------------------
    def add(self, rows) -> int:
        """Append rows [n, dim]; returns the (1-based) id of the first one."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if rows.ndim == 1:
            rows = rows[None, :]
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise _lib.MemexHipError(_lib.MX_EINVAL, f"expected [n, {self.dim}] rows, got {rows.shape}")
        first = ctypes.c_uint64(0)
        check(lib().mx_index_add(self._h, _ptr(rows), rows.shape[0], ctypes.byref(first)))
        return int(first.value)

    def add_device(self, t) -> int:
        """Append rows held in HBM (contiguous f32 [n, dim] tensor on this index's device)."""
        n = int(t.shape[0])
        first = ctypes.c_uint64(0)
        st = _caller_stream(t)
        if st is not None:
            self.wait_stream(st)
        check(lib().mx_index_add_device(self._h, ctypes.c_void_p(t.data_ptr()), n, ctypes.byref(first)))
        return int(first.value)

    def clear(self) -> None:
        check(lib().mx_index_clear(self._h))

    # -- search --------------------------------------------------------------------------
    def search(self, queries, k: int):
        """-> (ids u64 [B,k], scores f32 [B,k], dists f32 [B,k], n_found i32 [B])."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise _lib.MemexHipError(_lib.MX_EINVAL, f"expected [B, {self.dim}] queries, got {q.shape}")
        B = q.shape[0]
        ids = np.zeros((B, k), dtype=np.uint64)
        scores = np.zeros((B, k), dtype=np.float32)
        dists = np.zeros((B, k), dtype=np.float32)
        nf = np.zeros(B, dtype=np.int32)
        check(lib().mx_index_search(self._h, _ptr(q), B, int(k), _ptr(ids), _ptr(scores), _ptr(dists), _ptr(nf)))
        return ids, scores, dists, nf

    def search_device(self, q, k: int, ids, scores, dists, n_found) -> None:
        """All arguments are device tensors: q f32 [B,dim]; ids i64/u64 [B,k]; scores, dists f32 [B,k];
        n_found i32 [B].  Blocks until the results are in HBM."""
        B = int(q.shape[0])
        st = _caller_stream(q)
        if st is not None:  # q and the (possibly just zero-filled) outputs were produced on torch's stream
            self.wait_stream(st)
        check(lib().mx_index_search_device(self._h, ctypes.c_void_p(q.data_ptr()), B, int(k),
                                           ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(scores.data_ptr()),
                                           ctypes.c_void_p(dists.data_ptr()) if dists is not None else None,
                                           ctypes.c_void_p(n_found.data_ptr())))

    # -- persistence ---------------------------------------------------------------------
    def save(self, directory: str) -> None:
        check(lib().mx_index_save(self._h, str(directory).encode()))

    def load(self, directory: str) -> None:
        check(lib().mx_index_load(self._h, str(directory).encode()))

    @staticmethod
    def has_store(directory: str) -> bool:
        e = ctypes.c_int(0)
        check(lib().mx_index_has_store(str(directory).encode(), ctypes.byref(e)))
        return bool(e.value)

    @staticmethod
    def store_info(directory: str):
        """-> (dim, n_rows) of the persisted vector file."""
        d = ctypes.c_int(0)
        n = ctypes.c_uint64(0)
        check(lib().mx_index_store_info(str(directory).encode(), ctypes.byref(d), ctypes.byref(n)))
        return int(d.value), int(n.value)

    @staticmethod
    def remove_files(directory: str) -> None:
        check(lib().mx_index_remove_files(str(directory).encode()))


def merge_topk_device(device: int, ids, dists, out_ids, out_dists, out_scores) -> None:
    """ids/dists: device tensors [G,B,k] (gathered shard results) -> [B,k] global top-k."""
    G, B, k = (int(x) for x in ids.shape)
    check(lib().mx_topk_merge_device(int(device), ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(dists.data_ptr()),
                                     G, B, k, ctypes.c_void_p(out_ids.data_ptr()),
                                     ctypes.c_void_p(out_dists.data_ptr()),
                                     ctypes.c_void_p(out_scores.data_ptr()) if out_scores is not None else None))


def packed_result_block(B: int, k: int, device):
    """One contiguous device block [ids: B*k i64][dists: B*k f32] plus its two views: search results
    written through the views travel in ONE all-gather."""
    import torch
    block = torch.zeros((B * k * 12,), dtype=torch.uint8, device=device)
    ids = block[: B * k * 8].view(torch.int64).view(B, k)
    dists = block[B * k * 8:].view(torch.float32).view(B, k)
    return block, ids, dists


def merge_topk_packed_device(device: int, packed, G: int, B: int, k: int, out_ids, out_dists, out_scores,
                             stream: int | None = None) -> None:
    """packed: uint8 device tensor [G, B*k*12], shard blocks as laid out by packed_result_block.
    ``stream`` (raw hipStream_t): enqueue the merge there and return at once -- e.g. torch's current
    stream, right behind the all-gather; without it the call blocks until the merge is done."""
    args = (ctypes.c_void_p(packed.data_ptr()), int(G), int(B), int(k), ctypes.c_void_p(out_ids.data_ptr()),
            ctypes.c_void_p(out_dists.data_ptr()), ctypes.c_void_p(out_scores.data_ptr()) if out_scores is not None else None)
    if stream is None:
        check(lib().mx_topk_merge_packed_device(int(device), *args))
    else:
        check(lib().mx_topk_merge_packed_async(int(device), ctypes.c_void_p(stream), *args))
