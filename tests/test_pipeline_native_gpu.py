"""BASELINE.json configs[0] as far as it can run offline (no checkpoint, no vocab.txt, and the reference's
example document stays under /root/reference): the reference's ingest + query flow, end to end on the
HIP path with the NATIVE tokenizer --
    segment_text (embedding.rs:155-198)  ->  mx_tokenizer_segment
    model.encode tokenisation (:109)     ->  mx_tokenizer_encode_batch
    model.encode forward                 ->  mx_encoder_encode
    add_vectors (tasks.rs:59)            ->  HipFlatStore.bulk_insert -> mx_index_add (+ save)
    search (handlers.rs:72-81)           ->  encode_single -> mx_index_search
against the CPU path: `tokenizers` (the crate the reference calls) for windows and ids, the f64 encoder
oracle for vectors, the search oracle for hits.  Bars: windows and token ids identical; embeddings within
1e-3 cosine; hits on identical f32 vectors bit-exact; scores vs the all-CPU pipeline within 1e-3."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-3

WORDS = ("the of and to in a is that for it as was with be by on not he i this are or his from at which but have an had "
         "they you were their one all we can her has there been if more when will would who so no tax taxes economy "
         "jobs america american people congress year years work working families union state nation world president "
         "biden says said say about what does plan pay fair share billion million percent deficit health care cost "
         "costs lower drug prices energy climate future children school teachers union build built building made "
         "chips infrastructure roads bridges freedom democracy together finish job").split()
PIECES = ["s", "ed", "ing", "er", "ers", "ly", "ion", "ions", "al", "ity", "ment", "est", "un", "re", "in", "an", "en"]
PUNCT = list(".,;:!?'\"()-")


def _vocab(tmp_path):
    toks = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    toks += list("abcdefghijklmnopqrstuvwxyz0123456789") + ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789"]
    toks += PUNCT + sorted(set(WORDS)) + ["##" + p for p in PIECES] + ["cafe", "resume", "naive"]
    seen, out = set(), []
    for t in toks:
        if t not in seen:
            seen.add(t)
            out.append(t)
    p = tmp_path / "vocab.txt"
    p.write_text("\n".join(out) + "\n", encoding="utf-8")
    return str(p), out


def _document(rng, n_words):
    out = []
    for i in range(n_words):
        w = str(rng.choice(WORDS))
        r = rng.random()
        if r < 0.08:
            w = w.capitalize()
        elif r < 0.10:
            w = w.upper()
        elif r < 0.16:
            w = w + str(rng.choice(["s", "ed", "ing", "ers", "ly"]))
        elif r < 0.18:
            w = str(rng.choice(["Café", "résumé", "naïve", "2023", "1,000", "zzqx"]))
        out.append(w)
        if rng.random() < 0.12:
            out[-1] += str(rng.choice([".", ",", ";", "!", "?", " --", "'s", ":"]))
    return " ".join(out)


def test_cfg1_ingest_and_query_with_the_native_tokenizer(tmp_path, oracle, lib_built):
    from tokenizers import BertWordPieceTokenizer
    from memex_amd import embedding as E
    from memex_amd.storage import VectorData, evict_resident, get_vector_storage
    from memex_amd.weights import EncoderConfig, synthetic_weights
    from oracle import bert_oracle

    vocab_path, vocab = _vocab(tmp_path)
    cfg = EncoderConfig(layers=3, hidden=384, heads=12, ffn=1536, vocab=len(vocab), max_seq_length=128)   # L12-v2 truncation (App. A.1)
    w = synthetic_weights(cfg, 21)
    rng = np.random.default_rng(21)
    doc = _document(rng, 6000)
    query = "What does Biden say about taxes?"                         # README.md:104

    # ---- HIP path: embedder actor with the native tokenizer (a vocab.txt path selects it)
    mc = E.ModelConfig()                                               # default model, max_length 256, stride 86
    th, emb = E.SentenceEmbedder.spawn(mc, weights=w, tokenizer=vocab_path, encoder_config=cfg)
    segs = emb.encode(doc)                                             # tasks.rs:19
    qres = emb.encode_single(query)                                    # handlers.rs:72
    # the two callers themselves (memex_amd/tasks.py = tasks.rs:9-66 / handlers.rs:72-85 without the SQL): worker handle
    # ingests task 1, API handle searches; ids are the reference's v5 uuids
    from memex_amd import tasks as T
    uri2 = f"hnsw://{tmp_path / 'store2'}"
    written = T.process_embeddings(get_vector_storage(uri2, "test"), emb, 1, doc)
    hits = T.search_docs(get_vector_storage(uri2, "test"), emb, query, 3)
    emb.shutdown()
    assert [v._id for v in written] == [T.segment_uuid(T.document_uuid(1), i) for i in range(len(segs))]
    assert [v.text for v in written] == [s_.content for s_ in segs]

    # ---- CPU path, tokenisation: the `tokenizers` package with the reference's call sequence
    hf = E.HFTokenizerAdapter(BertWordPieceTokenizer(vocab_path, lowercase=True)._tokenizer)
    want_windows = hf.windows(doc, mc.max_length, mc.stride)
    assert [s.content for s in segs] == want_windows                   # same windows, same detokenised text
    assert len(segs) > 20
    ids, lens = hf.encode_batch(want_windows, cfg.max_seq_length)
    from memex_amd.tokenizer import WordPieceTokenizer
    nids, nlens = WordPieceTokenizer(vocab_path).encode_batch(want_windows, cfg.max_seq_length)
    np.testing.assert_array_equal(nlens, lens)
    np.testing.assert_array_equal(nids, ids)                           # same ids into the encoder

    # ---- CPU path, vectors: f64 oracle on those ids
    ref = bert_oracle.encode(w, cfg.as_dict(), ids, lens).astype(np.float32)
    vecs = np.asarray([s.vector for s in segs], dtype=np.float32)
    assert (1.0 - (vecs * ref).sum(1)).max() <= TOL
    qi, ql = hf.encode_batch([query], cfg.max_seq_length)
    qref = bert_oracle.encode(w, cfg.as_dict(), qi, ql).astype(np.float32)[0]
    q = np.asarray(qres.vector, dtype=np.float32)
    assert 1.0 - float(q @ qref) <= TOL

    # ---- store + search through the reference's surface (two handles, like worker and API)
    uri = f"hnsw://{tmp_path / 'store'}"
    get_vector_storage(uri, "test").add_vectors(
        [VectorData(_id=f"seg-{i}", document_id="doc", text=s.content, vector=s.vector, segment_id=i) for i, s in enumerate(segs)])
    got = get_vector_storage(uri, "test").search(q, 3)                 # README.md:104: limit 3
    oi, _, os_, _ = oracle.search(vecs, q, 3)                          # identical f32 vectors -> bit-exact
    assert [g[0] for g in got] == [f"seg-{int(i) - 1}" for i in oi[0]]
    np.testing.assert_array_equal(np.float32([g[1] for g in got]), os_[0])
    assert [h[0] for h in hits] == [written[int(i) - 1]._id for i in oi[0]]       # the callers' route: same neighbours, by uuid
    np.testing.assert_array_equal(np.float32([h[1] for h in hits]), os_[0])
    _, _, cpu_scores, _ = oracle.search(ref, qref, 3)                  # the all-CPU pipeline
    assert np.abs(np.float32([g[1] for g in got]) - cpu_scores[0]).max() <= TOL
    get_vector_storage(uri, "test").delete_collection()
    get_vector_storage(uri2, "test").delete_collection()
    evict_resident()
