"""Host-side mirror of memex's vector-store surface, served by the HIP flat index.

Same names, argument meaning and error behaviour as the reference
(lib/libmemex/src/storage/mod.rs:17-139, lib/libmemex/src/storage/local.rs:21-166) so the parity
tests read like the reference's own tests (local.rs:168-243):

=====================  =======================================================================
reference              here
=====================  =======================================================================
``VectorData``         :class:`VectorData`                               (mod.rs:17-28)
``VectorStoreError``   :class:`VectorStoreError` + one subclass per variant (mod.rs:31-48)
``VectorStore`` trait  :class:`VectorStore` ABC                          (mod.rs:55-66)
``HnswStore``          :class:`HipFlatStore` (exact search on the GPU)   (local.rs:21-166)
``VectorStorage``      :class:`VectorStorage` (mutex wrapper)            (mod.rs:69-93)
``get_vector_storage`` :func:`get_vector_storage` (``hnsw://`` and ``hip://`` URIs) (mod.rs:95-139)
=====================  =======================================================================

The reference's methods are ``async`` (tokio); here they are plain blocking calls -- the C ABI
underneath is synchronous and the Rust shim of INTEGRATION.md wraps it in ``async fn`` again.
"""
from __future__ import annotations

import abc
import json
import os
import threading
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple
from urllib.parse import urlparse

import numpy as np

from . import _lib
from .index import FlatIndex

META_FILE = "vectors.meta.json"  # local.rs:19 -- same file name and JSON shape as the reference


@dataclass
class VectorData:
    """mod.rs:17-28."""
    _id: str
    document_id: str
    text: str
    vector: Sequence[float]
    segment_id: int = 0

    def to_json(self) -> dict:
        """The serialisation boundary (the reference writes the vector as JSON into the `embeddings` table:
        `vector.into()`, worker/tasks.rs:42-48): `vector` may be a numpy float32 row, which `json.dumps` refuses."""
        return {"_id": self._id, "document_id": self.document_id, "text": self.text,
                "vector": [float(x) for x in self.vector], "segment_id": int(self.segment_id)}


class VectorStoreError(Exception):
    """mod.rs:31-48; subclasses below are the enum variants."""


class ConnectionError_(VectorStoreError):
    pass


class DeleteError(VectorStoreError):
    pass


class FileIOError(VectorStoreError):
    pass


class InsertionError(VectorStoreError):
    pass


class SearchError(VectorStoreError):
    pass


class SerdeError(VectorStoreError):
    pass


class SaveError(VectorStoreError):
    pass


class Unsupported(VectorStoreError):
    pass


_STATUS_TO_ERROR = {
    _lib.MX_EINVAL: InsertionError,
    _lib.MX_EDEVICE: ConnectionError_,
    _lib.MX_EINSERT: InsertionError,
    _lib.MX_ESEARCH: SearchError,
    _lib.MX_EIO: FileIOError,
    _lib.MX_EUNSUPPORTED: Unsupported,
    _lib.MX_ENOMEM: InsertionError,
}


def _raise_from(e: _lib.MemexHipError, default=None):
    cls = _STATUS_TO_ERROR.get(e.code, default or VectorStoreError)
    if default is not None and e.code == _lib.MX_EINVAL:
        cls = default
    raise cls(e.msg) from e


VectorSearchResult = Tuple[str, float]  # (doc_id, score), mod.rs:51


class VectorStore(abc.ABC):
    """mod.rs:55-66."""

    @abc.abstractmethod
    def delete(self, id: str) -> None: ...

    @abc.abstractmethod
    def delete_all(self) -> None: ...

    @abc.abstractmethod
    def bulk_insert(self, data: Sequence[VectorData]) -> None: ...

    @abc.abstractmethod
    def insert(self, data: VectorData) -> None: ...

    @abc.abstractmethod
    def search(self, vec: Sequence[float], limit: int) -> List[VectorSearchResult]: ...


# Process-wide registry of resident stores, keyed by (real storage path, device).  The reference builds
# a store per request / per task (handlers.rs:61-63, worker/lib.rs:190) and pays a full reload each
# time (storage/mod.rs:115-116); here a collection that is already in HBM is attached in O(1): the
# SAME store object (GPU index + id map) is handed out as long as the files it last wrote / read are
# unchanged on disk.  (The Rust shim keeps the equivalent `HashMap<PathBuf, Arc<Mutex<HipFlatStore>>>`,
# INTEGRATION.md.)
_RESIDENT: Dict[tuple, "HipFlatStore"] = {}
_RESIDENT_MU = threading.Lock()


def _file_sig(path: str):
    try:
        st = os.stat(path)
        return (st.st_mtime_ns, st.st_size)
    except OSError:
        return None


def evict_resident(storage_path: str | None = None) -> None:
    """Drop resident stores (all, or the one for ``storage_path``): frees their HBM once unused."""
    with _RESIDENT_MU:
        for key in [k for k in _RESIDENT if storage_path is None or k[0] == os.path.realpath(str(storage_path))]:
            st = _RESIDENT.pop(key)
            with st._lock:      # a search in flight holds its own reference: the handle closes when that one goes
                st._index = None


@dataclass
class HipFlatStore(VectorStore):
    """Drop-in for ``HnswStore``: same id map, same score formula, exact search on the GPU."""
    storage_path: str
    device: int = 0
    devices: Sequence[int] | None = None   # several GPUs: the in-library sharded index (mx_index_open_sharded)
    _id_map: Dict[int, str] = field(default_factory=dict)
    _index: FlatIndex | None = None
    _dim: int | None = None
    _lock: threading.RLock = field(default_factory=threading.RLock, repr=False, compare=False)
    _meta_sig: tuple | None = None          # (mtime_ns, size) of vectors.meta.json as last written / read
    _meta_ids: int = 0                      # ids that file holds

    # -- construction (local.rs:95-141) ----------------------------------------------------
    @classmethod
    def new(cls, storage_path: str, device: int = 0, devices: Sequence[int] | None = None) -> "HipFlatStore":
        store = cls(storage_path=str(storage_path), device=device, devices=devices)
        with _RESIDENT_MU:
            # a store this one replaces is NOT closed under its holders: earlier requests may still be inside
            # search() on it; its handle on the keyed GPU index goes away with its last reference
            _RESIDENT[store._rkey()] = store
        return store

    def _rkey(self) -> tuple:
        return (os.path.realpath(self.storage_path), int(self.device), tuple(int(d) for d in self.devices) if self.devices else None)

    @staticmethod
    def has_store(store_path: str) -> bool:
        return os.path.exists(os.path.join(str(store_path), META_FILE))  # local.rs:110-113

    @classmethod
    def load(cls, store_path: str, device: int = 0, devices: Sequence[int] | None = None) -> "HipFlatStore":
        store_path = str(store_path)
        meta = os.path.join(store_path, META_FILE)
        probe = cls(storage_path=store_path, device=device, devices=devices)
        with _RESIDENT_MU:
            res = _RESIDENT.get(probe._rkey())
        if res is not None:
            with res._lock:
                if res._meta_sig is not None and res._meta_sig == _file_sig(meta) and len(res._id_map) == res._meta_ids:
                    if not res._id_map:
                        return res
                    try:
                        res._index.load(store_path)      # O(1) when vectors.mxflat is what this index last wrote / read
                    except _lib.MemexHipError as e:
                        _raise_from(e)
                    if len(res._index) == len(res._id_map):
                        return res
        try:
            with open(meta, "r", encoding="utf-8") as f:
                raw = json.load(f)
        except OSError as e:
            raise FileIOError(str(e)) from e
        except ValueError as e:
            raise SerdeError(str(e)) from e
        store = probe
        try:
            store._id_map = {int(k): str(v) for k, v in raw.items()}
        except (AttributeError, ValueError) as e:
            raise SerdeError(str(e)) from e
        store._meta_sig = _file_sig(meta)
        store._meta_ids = len(store._id_map)
        if store._id_map:
            try:
                if not FlatIndex.has_store(store_path):
                    raise FileIOError(f"{store_path}: vector file missing")
                dim, _ = FlatIndex.store_info(store_path)
                store._open(dim)
                store._index.load(store_path)
            except _lib.MemexHipError as e:
                _raise_from(e)
            if len(store._index) != len(store._id_map):
                raise FileIOError(f"{store_path}: {len(store._index)} vectors vs {len(store._id_map)} ids")
        with _RESIDENT_MU:
            _RESIDENT[store._rkey()] = store     # (a replaced store keeps its handle for whoever still uses it)
        return store

    def save(self, store_path: str | None = None) -> None:
        """local.rs:143-165: vectors + ``vectors.meta.json`` (``{"<usize>": "<_id>"}``).  The reference
        calls this after EVERY insert (local.rs:67) and rewrites both files; here a save into the
        store's own directory appends: the index adds the new rows to ``vectors.mxflat``
        (``mx_index_save``) and the JSON object gets the new ``"id": "_id"`` pairs spliced in before
        its closing brace -- same file format, O(new rows) per insert."""
        store_path = str(store_path or self.storage_path)
        with self._lock:
            try:
                os.makedirs(store_path, exist_ok=True)
                if self._index is not None:
                    self._index.save(store_path)
                meta = os.path.join(store_path, META_FILE)
                own = os.path.realpath(store_path) == os.path.realpath(self.storage_path)
                n = len(self._id_map)
                appended = False
                if own and self._meta_sig is not None and self._meta_sig == _file_sig(meta) and 0 < self._meta_ids <= n:
                    appended = self._meta_ids == n or self._append_meta(meta, n)
                if not appended:
                    # full rewrite through a temporary file: the vectors are already on disk, so the id map must
                    # never be left shorter than them (also the way out of a file the splice does not recognise,
                    # e.g. one that a hand edit ended with a newline)
                    tmp = meta + ".tmp"
                    with open(tmp, "w", encoding="utf-8") as f:
                        json.dump({str(k): v for k, v in self._id_map.items()}, f)
                    os.replace(tmp, meta)
                if own:
                    self._meta_sig = _file_sig(meta)
                    self._meta_ids = n
            except _lib.MemexHipError as e:
                self._meta_sig = None
                raise SaveError(e.msg) from e
            except OSError as e:
                self._meta_sig = None
                raise FileIOError(str(e)) from e

    def _append_meta(self, meta: str, n: int) -> bool:
        """Splice ids _meta_ids+1 .. n in front of the closing brace of vectors.meta.json.  False = the file does
        not end the way this store left it (nothing written): the caller rewrites it."""
        tail = ",".join(f"{json.dumps(str(i))}: {json.dumps(self._id_map[i])}" for i in range(self._meta_ids + 1, n + 1))
        try:
            with open(meta, "r+b") as f:
                f.seek(-1, os.SEEK_END)
                if f.read(1) != b"}":
                    return False
                f.seek(-1, os.SEEK_END)
                f.write((", " + tail + "}").encode("utf-8"))
            return True
        except OSError:
            return False

    def _open(self, dim: int) -> None:
        # keyed by the collection's path: every handle on this collection shares ONE resident GPU index
        key = f"{os.path.realpath(self.storage_path)}@{'+'.join(map(str, self.devices)) if self.devices else self.device}"
        try:
            self._index = FlatIndex(dim, key=key, device=self.device, devices=self.devices)
        except _lib.MemexHipError as e:
            _raise_from(e)
        self._dim = dim

    # -- VectorStore (local.rs:27-92) -------------------------------------------------------
    def delete(self, id: str) -> None:
        # local.rs:29-32 is `unimplemented!()` (a panic); same contract, as an exception
        raise NotImplementedError("single-point delete is not supported (reference: unimplemented!())")

    def delete_all(self) -> None:
        with self._lock:
            for name in (META_FILE,):
                p = os.path.join(self.storage_path, name)
                if os.path.exists(p):
                    os.remove(p)
            try:
                FlatIndex.remove_files(self.storage_path)
                if self._index is not None:
                    self._index.clear()
            except _lib.MemexHipError as e:
                raise DeleteError(e.msg) from e
            self._id_map.clear()
            self._meta_sig = None
            self._meta_ids = 0

    def bulk_insert(self, data: Sequence[VectorData]) -> None:
        """local.rs:55-69 semantics (ids in order, store persisted before returning), one device
        transfer and one incremental save instead of a save per vector."""
        if not data:
            return
        rows = np.asarray([np.asarray(d.vector, dtype=np.float32) for d in data], dtype=np.float32)
        if rows.ndim != 2:
            raise InsertionError("vectors of one bulk_insert must share a dimension")
        with self._lock:
            if self._index is None:
                self._open(rows.shape[1])
                if len(self._index) != len(self._id_map):   # a stale resident index under this key
                    self._index.clear()
            if rows.shape[1] != self._dim:
                raise InsertionError(f"vector dimension {rows.shape[1]} != store dimension {self._dim}")
            try:
                first = self._index.add(rows)
            except _lib.MemexHipError as e:
                _raise_from(e, InsertionError)
            next_id = len(self._id_map) + 1  # local.rs:63
            assert first == next_id, (first, next_id)
            for i, d in enumerate(data):
                self._id_map[next_id + i] = str(d._id)
            try:
                self.save()  # local.rs:67 `let _ = self.save(..)`: errors are ignored there too
            except VectorStoreError:
                pass

    def insert(self, data: VectorData) -> None:
        self.bulk_insert([data])

    def search(self, vec: Sequence[float], limit: int) -> List[VectorSearchResult]:
        with self._lock:
            idx = self._index        # own reference: the store may be evicted / replaced while the GPU works
        if idx is None or limit <= 0:
            return []
        q = np.asarray(vec, dtype=np.float32)
        if q.shape != (self._dim,):
            raise SearchError(f"query dimension {q.shape} != store dimension {self._dim}")
        try:
            ids, scores, _, nf = idx.search(q, int(limit))      # not under the lock: concurrent callers are combined
        except _lib.MemexHipError as e:
            _raise_from(e, SearchError)
        out: List[VectorSearchResult] = []
        with self._lock:
            for j in range(int(nf[0])):
                d_id = int(ids[0, j])
                if d_id not in self._id_map:  # local.rs:80-83 panics here; we raise
                    raise SearchError("Internal inconsistency. Id from vector store not mapped.")
                out.append((self._id_map[d_id], float(scores[0, j])))
        return out


class VectorStorage:
    """mod.rs:69-93: a mutex around a ``VectorStore``."""

    def __init__(self, client: VectorStore):
        self.client = client
        self._mu = threading.Lock()

    def add_vectors(self, points: Sequence[VectorData]) -> None:
        with self._mu:
            self.client.bulk_insert(points)

    def delete_collection(self) -> None:
        with self._mu:
            self.client.delete_all()

    def search(self, query: Sequence[float], limit: int) -> List[VectorSearchResult]:
        with self._mu:
            return self.client.search(query, limit)


def get_vector_storage(uri: str, collection: str, device: int = 0, devices: Sequence[int] | None = None) -> VectorStorage:
    """mod.rs:95-139.  ``hnsw://<dir>`` (the reference's file backend, now served from HBM) and
    ``hip://<dir>`` select the GPU store; collections are folders under ``<dir>``.  Called per
    request like the reference's; a collection that is already resident is attached, not reloaded."""
    try:
        scheme = urlparse(uri).scheme
    except ValueError:
        raise Unsupported(uri)
    if not scheme:
        raise Unsupported(uri)
    if scheme in ("hnsw", "hip"):
        storage = os.path.join(uri[len(scheme) + 3:], collection)
        try:
            os.makedirs(storage, exist_ok=True)
        except OSError as e:
            raise FileIOError(str(e)) from e
        if HipFlatStore.has_store(storage):
            store = HipFlatStore.load(storage, device, devices)
        else:
            probe = HipFlatStore(storage_path=storage, device=device, devices=devices)
            with _RESIDENT_MU:
                res = _RESIDENT.get(probe._rkey())
            # an empty resident store (created by an earlier request, nothing inserted yet) is reused
            store = res if res is not None and not res._id_map else HipFlatStore.new(storage, device, devices)
        return VectorStorage(store)
    # opensearch+https:// is a remote-service client in the reference (mod.rs:122-133): out of scope
    raise Unsupported(uri)
