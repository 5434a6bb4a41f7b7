"""Where a document's ingest time goes on the host mirror (not a test): the steps of process_embeddings, one document at a time.
usage: gpu_text_ingest_profile.py [documents]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from memex_amd import tasks, weights as W
from memex_amd.encoder import Encoder
from memex_amd.storage import VectorData, get_vector_storage
from memex_amd.tokenizer import WordPieceTokenizer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
vocab, docs = bench._synthetic_vocab_and_docs(n, 42_000)
tok = WordPieceTokenizer(vocab)
cfg = W.ALL_MINILM_L12_V2
enc = Encoder(cfg, W.synthetic_weights(cfg, 0))
acc = {}
def lap(name, t0):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
with tempfile.TemporaryDirectory() as td:
    client = get_vector_storage("hip://" + td, "prof")
    for i, d in enumerate(docs):
        t0 = time.perf_counter(); segs = tok.windows(d, 256, 86); lap("segment", t0)
        t0 = time.perf_counter(); ids, lens = tok.encode_batch(segs, cfg.max_seq_length); lap("encode_batch (tokenise windows)", t0)
        t0 = time.perf_counter(); vecs = enc.encode(ids, lens); lap("encoder", t0)
        t0 = time.perf_counter()
        doc = tasks.document_uuid(i)
        vd = [VectorData(_id=tasks.segment_uuid(doc, k), document_id=doc, text=s, vector=v, segment_id=k) for k, (s, v) in enumerate(zip(segs, vecs))]
        lap("uuid5 + VectorData", t0)
        t0 = time.perf_counter(); client.add_vectors(vd); lap("add_vectors (rows -> GPU index, incremental save)", t0)
        if i == 0:
            acc.clear()   # warm-up document
    client.delete_collection()
tot = sum(acc.values())
print(f"{n - 1} documents, {tot / (n - 1) * 1e3:.2f} ms per document one at a time ({len(segs)} windows in the last one)")
for k, v in acc.items():
    print(f"  {k:55s} {v / (n - 1) * 1e3:7.3f} ms")
