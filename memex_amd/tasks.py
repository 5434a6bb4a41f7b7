"""The two callers of the hot path, as far as the path goes (no SQL, no HTTP): what the worker does with a document and what
the API does with a query.

=================================  ==========================================================================
reference                          here
=================================  ==========================================================================
``process_embeddings``             :func:`process_embeddings` -- embed the document's windows, name every segment
(lib/worker/src/tasks.rs:9-66)     ``v5(NAMESPACE, "{doc_uuid}-{idx}")`` with ``doc_uuid = v5(NAMESPACE, task_id)``
                                   (db/document.rs:73-74), hand them to ``add_vectors``; returns the ``VectorData`` the
                                   reference also writes to its ``embeddings`` table
``handle_search_docs``             :func:`search_docs` -- ``encode_single(query)`` -> ``search(vector, limit)`` ->
(lib/api/.../handlers.rs:55-109)   ``(segment _id, score)`` pairs (the handler then looks each _id up in SQL)
=================================  ==========================================================================

The ids are the reference's own (RFC 4122 v5 over the same namespace), so a collection filled through this mirror lines up
with the rows a memex worker would have written for the same task ids.
"""
from __future__ import annotations

import uuid
from typing import List, Sequence, Tuple

from .storage import VectorData

NAMESPACE = uuid.UUID("5fdfe40a-de2c-11ed-bfa7-00155deae876")   # lib/libmemex/src/lib.rs:6


def document_uuid(task_id: int) -> str:
    """db/document.rs:74: ``Uuid::new_v5(&NAMESPACE, task.id.to_string().as_bytes())``."""
    return str(uuid.uuid5(NAMESPACE, str(task_id)))


def segment_uuid(doc_uuid: str, idx: int) -> str:
    """tasks.rs:36-40: ``Uuid::new_v5(&NAMESPACE, format!("{doc_uuid}-{idx}").as_bytes())``."""
    return str(uuid.uuid5(NAMESPACE, f"{doc_uuid}-{idx}"))


def process_embeddings(client, embedder, task_id: int, content: str) -> List[VectorData]:
    """tasks.rs:9-66 without the SQL: ``embedder.encode(content)`` -> one ``VectorData`` per window -> ``client.add_vectors``.
    Like the reference, a failing ``add_vectors`` is the caller's to log (the exception propagates here)."""
    embeddings = embedder.encode(content)                                  # :19
    doc = document_uuid(task_id)                                           # :28 (document::ActiveModel::from_task)
    vectors = [VectorData(_id=segment_uuid(doc, idx), document_id=doc, text=e.content, vector=e.vector, segment_id=idx)
               for idx, e in enumerate(embeddings)]                        # :34-56
    client.add_vectors(vectors)                                            # :59
    return vectors


def search_docs(client, embedder, query: str, limit: int = 10) -> List[Tuple[str, float]]:
    """handlers.rs:72-85: embed the query, search; ``ValueError("Invalid query")`` where the handler rejects (:74-78)."""
    res = embedder.encode_single(query)
    if res is None:
        raise ValueError("Invalid query")
    return client.search(res.vector, limit)
