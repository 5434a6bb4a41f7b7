"""Host-side logic that needs no GPU: URI factory, error mapping, stage planning, sharding."""
import numpy as np
import pytest

from memex_amd import storage
from memex_amd.sharded import partition


def test_get_vector_storage_rejects_unknown_schemes(tmp_path):
    # reference: storage/mod.rs:99-102,134-136 -> VectorStoreError::Unsupported
    for uri in ("", "not a uri", "qdrant://x", "opensearch+https://admin:admin@localhost:9200", "file:///tmp/x"):
        with pytest.raises(storage.Unsupported):
            storage.get_vector_storage(uri, "c")


def test_vector_data_fields_match_reference():
    v = storage.VectorData(_id="a", document_id="d", text="t", vector=[0.0], segment_id=3)
    assert (v._id, v.document_id, v.text, v.segment_id) == ("a", "d", "t", 3)   # mod.rs:17-28


def test_vector_data_serialises_with_a_numpy_row():
    """The embedder hands out numpy float32 rows (embedding.py::EmbeddingResult); the reference serialises the vector as JSON
    (`vector.into()`, worker/tasks.rs:42-48): `to_json()` is that boundary."""
    import json
    import numpy as np
    from memex_amd.storage import VectorData
    v = np.arange(6, dtype=np.float32) / 3
    d = VectorData(_id="a", document_id="d", text="t", vector=v, segment_id=np.int64(2))
    back = json.loads(json.dumps(d.to_json()))
    assert back["segment_id"] == 2 and np.array_equal(np.array(back["vector"], dtype=np.float32), v)


def test_store_without_rows_needs_no_device(tmp_path):
    s = storage.HipFlatStore.new(str(tmp_path))
    assert s.search([0.1, 0.2], 3) == []                 # empty store: nothing to search, no GPU touched
    with pytest.raises(NotImplementedError):
        s.delete("x")                                    # reference: unimplemented!() (local.rs:29-32)
    assert not storage.HipFlatStore.has_store(str(tmp_path))
    s.save()
    assert storage.HipFlatStore.has_store(str(tmp_path))   # vectors.meta.json exists (local.rs:110-113)
    s2 = storage.HipFlatStore.load(str(tmp_path))
    assert s2._id_map == {}
    s.delete_all()
    assert not storage.HipFlatStore.has_store(str(tmp_path))


def test_load_errors_map_to_reference_variants(tmp_path):
    with pytest.raises(storage.FileIOError):
        storage.HipFlatStore.load(str(tmp_path / "missing"))
    (tmp_path / storage.META_FILE).write_text("{not json")
    with pytest.raises(storage.SerdeError):
        storage.HipFlatStore.load(str(tmp_path))
    (tmp_path / storage.META_FILE).write_text('{"1": "a"}')      # ids but no vector file
    with pytest.raises(storage.FileIOError):
        storage.HipFlatStore.load(str(tmp_path))


def test_partition_covers_all_rows():
    for n, w in ((10_000_000, 8), (10, 3), (7, 8), (0, 2)):
        p = partition(n, w)
        assert p[0][0] == 0 and p[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(p, p[1:]))
        sizes = [b - a for a, b in p]
        assert max(sizes) - min(sizes) <= 1


def test_every_supported_model_type_has_an_encoder_config():
    """embedding.rs:25-33,159: the sentence-transformer checkpoints whose encoder this path covers."""
    from memex_amd import embedding as E
    from memex_amd import weights as W
    assert E._ENCODER_CONFIGS[E.EmbeddingsModelType.AllMiniLmL12V2] is W.ALL_MINILM_L12_V2     # the default (embedding.rs:67)
    assert E._ENCODER_CONFIGS[E.EmbeddingsModelType.AllMiniLmL6V2] is W.ALL_MINILM_L6_V2
    rb = E._ENCODER_CONFIGS[E.EmbeddingsModelType.AllDistilrobertaV1]
    assert (rb.layers, rb.hidden, rb.pos_offset, rb.type_vocab, rb.max_pos) == (6, 768, 2, 1, 514)
    assert rb.max_seq_length + rb.pos_offset <= rb.max_pos


def test_meta_save_appends_and_falls_back_to_a_rewrite(tmp_path):
    """save() splices new ids in front of the closing brace of vectors.meta.json (O(new ids) per insert, the
    reference saves after every insert: local.rs:67); a file that does not end the way the store left it --
    e.g. a hand edit that added a newline -- is rewritten in full instead of leaving fewer ids than vectors."""
    import json
    import os
    s = storage.HipFlatStore.new(str(tmp_path))
    s._id_map = {1: "a", 2: "b"}
    s.save()
    meta = tmp_path / storage.META_FILE
    assert json.loads(meta.read_text()) == {"1": "a", "2": "b"}
    s._id_map[3] = 'c "quoted"'
    s.save()                                               # spliced
    assert json.loads(meta.read_text()) == {"1": "a", "2": "b", "3": 'c "quoted"'}
    meta.write_text(meta.read_text() + "\n")               # external edit: trailing newline, still valid JSON
    s._meta_sig = storage._file_sig(str(meta))             # (as load() would record it)
    s._id_map[4] = "d"
    s.save()                                               # no '}' at the end -> full rewrite, nothing lost
    assert json.loads(meta.read_text()) == {"1": "a", "2": "b", "3": 'c "quoted"', "4": "d"}
    assert not os.path.exists(str(meta) + ".tmp")
    s._id_map[5] = "e"
    s.save()                                               # and the splice works again afterwards
    assert json.loads(meta.read_text())["5"] == "e"


def test_resident_registry_is_keyed_by_device_set(tmp_path):
    a = storage.HipFlatStore(storage_path=str(tmp_path), device=0)
    b = storage.HipFlatStore(storage_path=str(tmp_path), device=0, devices=[0, 1])
    c = storage.HipFlatStore(storage_path=str(tmp_path), device=0, devices=[0, 1])
    assert a._rkey() != b._rkey() and b._rkey() == c._rkey()


def test_reference_segment_ids():
    """The worker names a segment v5(NAMESPACE, "{doc_uuid}-{idx}") with doc_uuid = v5(NAMESPACE, task_id)
    (tasks.rs:36-40, db/document.rs:73-74, lib.rs:6).  RFC 4122 v5 is SHA-1 based: checked here against an independent
    hashlib computation, so the ids do not depend on the uuid module's own implementation."""
    import hashlib
    import uuid
    from memex_amd import tasks as T
    ns = uuid.UUID("5fdfe40a-de2c-11ed-bfa7-00155deae876")
    assert T.NAMESPACE == ns

    def v5(name: str) -> str:
        h = bytearray(hashlib.sha1(ns.bytes + name.encode()).digest()[:16])
        h[6] = (h[6] & 0x0F) | 0x50
        h[8] = (h[8] & 0x3F) | 0x80
        return str(uuid.UUID(bytes=bytes(h)))

    for task_id in (1, 42, 10 ** 12):
        doc = T.document_uuid(task_id)
        assert doc == v5(str(task_id))
        for idx in (0, 1, 54):
            assert T.segment_uuid(doc, idx) == v5(f"{doc}-{idx}")
    assert T.document_uuid(1) != T.document_uuid(2) and len(T.document_uuid(1)) == 36


def test_process_embeddings_and_search_docs_call_sequence():
    """tasks.rs:9-66 / handlers.rs:72-85 without the SQL: what reaches add_vectors / search, with stand-in client and embedder."""
    from memex_amd import tasks as T
    from memex_amd.embedding import EmbeddingResult

    class Emb:
        def encode(self, text):
            return [EmbeddingResult(content=f"w{i}", vector=[float(i), 1.0]) for i in range(3)]

        def encode_single(self, text):
            return None if not text else EmbeddingResult(content=text, vector=[1.0, 0.0])

    class Client:
        def __init__(self):
            self.added, self.queries = [], []

        def add_vectors(self, pts):
            self.added += list(pts)

        def search(self, vec, limit):
            self.queries.append((list(vec), limit))
            return [(p._id, 0.5) for p in self.added[:limit]]

    c = Client()
    out = T.process_embeddings(c, Emb(), 7, "some document")
    doc = T.document_uuid(7)
    assert [v._id for v in c.added] == [T.segment_uuid(doc, i) for i in range(3)] and out == c.added
    assert all(v.document_id == doc and v.segment_id == i and v.text == f"w{i}" for i, v in enumerate(c.added))
    assert T.search_docs(c, Emb(), "what about taxes?", 2) == [(c.added[0]._id, 0.5), (c.added[1]._id, 0.5)]
    assert c.queries == [([1.0, 0.0], 2)]
    import pytest
    with pytest.raises(ValueError):
        T.search_docs(c, Emb(), "")
