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


@dataclass
class HipFlatStore(VectorStore):
    """Drop-in for ``HnswStore``: same id map, same score formula, exact search on the GPU."""
    storage_path: str
    device: int = 0
    _id_map: Dict[int, str] = field(default_factory=dict)
    _index: FlatIndex | None = None
    _dim: int | None = None

    # -- construction (local.rs:95-141) ----------------------------------------------------
    @classmethod
    def new(cls, storage_path: str, device: int = 0) -> "HipFlatStore":
        return cls(storage_path=str(storage_path), device=device)

    @staticmethod
    def has_store(store_path: str) -> bool:
        return os.path.exists(os.path.join(str(store_path), META_FILE))  # local.rs:110-113

    @classmethod
    def load(cls, store_path: str, device: int = 0) -> "HipFlatStore":
        store_path = str(store_path)
        meta = os.path.join(store_path, META_FILE)
        try:
            with open(meta, "r", encoding="utf-8") as f:
                raw = json.load(f)
        except OSError as e:
            raise FileIOError(str(e)) from e
        except ValueError as e:
            raise SerdeError(str(e)) from e
        store = cls(storage_path=store_path, device=device)
        try:
            store._id_map = {int(k): str(v) for k, v in raw.items()}
        except (AttributeError, ValueError) as e:
            raise SerdeError(str(e)) from e
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
        return store

    def save(self, store_path: str | None = None) -> None:
        """local.rs:143-165: vectors + ``vectors.meta.json`` (``{"<usize>": "<_id>"}``)."""
        store_path = str(store_path or self.storage_path)
        try:
            os.makedirs(store_path, exist_ok=True)
            if self._index is not None:
                self._index.save(store_path)
            doc = {str(k): v for k, v in self._id_map.items()}
            with open(os.path.join(store_path, META_FILE), "w", encoding="utf-8") as f:
                json.dump(doc, f)
        except _lib.MemexHipError as e:
            raise SaveError(e.msg) from e
        except OSError as e:
            raise FileIOError(str(e)) from e

    def _open(self, dim: int) -> None:
        try:
            self._index = FlatIndex(dim, key=None, device=self.device)
        except _lib.MemexHipError as e:
            _raise_from(e)
        self._dim = dim

    # -- VectorStore (local.rs:27-92) -------------------------------------------------------
    def delete(self, id: str) -> None:
        # local.rs:29-32 is `unimplemented!()` (a panic); same contract, as an exception
        raise NotImplementedError("single-point delete is not supported (reference: unimplemented!())")

    def delete_all(self) -> None:
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

    def bulk_insert(self, data: Sequence[VectorData]) -> None:
        """local.rs:55-60 semantics (ids in order), one device transfer instead of a loop."""
        if not data:
            return
        rows = np.asarray([np.asarray(d.vector, dtype=np.float32) for d in data], dtype=np.float32)
        if rows.ndim != 2:
            raise InsertionError("vectors of one bulk_insert must share a dimension")
        if self._index is None:
            self._open(rows.shape[1])
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

    def insert(self, data: VectorData) -> None:
        self.bulk_insert([data])

    def search(self, vec: Sequence[float], limit: int) -> List[VectorSearchResult]:
        if self._index is None or limit <= 0:
            return []
        q = np.asarray(vec, dtype=np.float32)
        if q.shape != (self._dim,):
            raise SearchError(f"query dimension {q.shape} != store dimension {self._dim}")
        try:
            ids, scores, _, nf = self._index.search(q, int(limit))
        except _lib.MemexHipError as e:
            _raise_from(e, SearchError)
        out: List[VectorSearchResult] = []
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


def get_vector_storage(uri: str, collection: str, device: int = 0) -> VectorStorage:
    """mod.rs:95-139.  ``hnsw://<dir>`` (the reference's file backend, now served from HBM) and
    ``hip://<dir>`` select the GPU store; collections are folders under ``<dir>``."""
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
        store = HipFlatStore.load(storage, device) if HipFlatStore.has_store(storage) else HipFlatStore.new(storage, device)
        return VectorStorage(store)
    # opensearch+https:// is a remote-service client in the reference (mod.rs:122-133): out of scope
    raise Unsupported(uri)
