"""GPU parity of the HIP encoder (through the C ABI) against the oracle and the golden vectors.
Tolerance (north_star): cosine within 1e-3 of the f32/f64 CPU path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-3


class _DebugKeys:
    """MEMEX_HIP_DEBUG="key=value,key=value" (memex_amd/csrc/mx_debug.h: which of two kernel forms runs) as a dict bound to a
    test's monkeypatch: the library reads the string on every use, so a test may switch inside one process."""

    def __init__(self, monkeypatch):
        self.mp, self.kv = monkeypatch, {}

    def _flush(self):
        self.mp.setenv("MEMEX_HIP_DEBUG", ",".join(f"{k}={v}" for k, v in self.kv.items()))

    def set(self, key, value):
        self.kv[key] = str(value)
        self._flush()

    def unset(self, key):
        self.kv.pop(key, None)
        self._flush()


def _dbg(monkeypatch):
    if not hasattr(monkeypatch, "_mx_debug_keys"):
        monkeypatch._mx_debug_keys = _DebugKeys(monkeypatch)
    return monkeypatch._mx_debug_keys


def _cos(a, b):
    return (a * b).sum(1) / np.linalg.norm(a, axis=1) / np.linalg.norm(b, axis=1)


def test_encoder_vs_transformers_golden(lib_built):
    from memex_amd.encoder import Encoder
    from test_encoder_oracle import golden_cases
    for name, cfg, w, ids, lens, gold in golden_cases():
        with Encoder(cfg, w) as enc:
            out = enc.encode(ids, lens)
        assert np.isfinite(out).all(), name
        assert (1.0 - _cos(out.astype(np.float64), gold)).max() <= TOL, name
        np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)


@pytest.mark.parametrize("kw,B,S,seed", [
    (dict(layers=6, hidden=384, heads=12, ffn=1536, vocab=3000), 6, 256, 1),        # all-MiniLM-L6-v2 shape
    (dict(layers=12, hidden=384, heads=12, ffn=1536, vocab=3000), 3, 128, 2),       # all-MiniLM-L12-v2 shape (default)
    (dict(layers=6, hidden=384, heads=12, ffn=1536, vocab=3000), 2, 512, 3),        # max positions
    (dict(layers=12, hidden=768, heads=12, ffn=3072, vocab=3000, pooling="cls"), 3, 160, 4),  # bge-base shape
    (dict(layers=2, hidden=384, heads=12, ffn=1536, vocab=3000, normalize=False), 5, 33, 5),
    (dict(layers=2, hidden=768, heads=12, ffn=3072, vocab=3000), 9, 77, 6),          # mean pooling at H=768
    (dict(layers=6, hidden=768, heads=12, ffn=3072, vocab=3000, max_pos=514, type_vocab=1, ln_eps=1e-5, pos_offset=2),
     3, 512, 7),                                                                     # all-distilroberta-v1 shape (embedding.rs:29)
    (dict(layers=6, hidden=384, heads=12, ffn=1536, vocab=3000, precision="bf16x3"), 6, 256, 1),   # the split-operand mode
    (dict(layers=12, hidden=768, heads=12, ffn=3072, vocab=3000, pooling="cls", precision="bf16x3"), 3, 160, 4),
    (dict(layers=2, hidden=768, heads=12, ffn=3072, vocab=3000, normalize=False, precision="bf16x3"), 5, 512, 8),
    (dict(layers=3, hidden=768, heads=12, ffn=3072, vocab=3000, max_pos=514, type_vocab=1, ln_eps=1e-5, pos_offset=2,
          precision="bf16x3"), 4, 300, 9),                                           # RoBERTa-style tables in the split-operand mode
    (dict(layers=2, hidden=384, heads=6, ffn=768, vocab=3000, pooling="cls", precision="bf16x3"), 7, 90, 10),   # head dim 64 at hidden 384
    (dict(layers=6, hidden=384, heads=12, ffn=1536, vocab=3000, precision="mixed"), 6, 256, 1),    # the mixed mode: MLP on two fp16 products
    (dict(layers=12, hidden=768, heads=12, ffn=3072, vocab=3000, pooling="cls", precision="mixed"), 3, 160, 4),
    (dict(layers=2, hidden=768, heads=12, ffn=3072, vocab=3000, normalize=False, precision="mixed"), 5, 512, 8),
    (dict(layers=6, hidden=384, heads=12, ffn=1536, vocab=3000, precision="mixed1"), 6, 256, 1),   # ... on ONE fp16 product
    (dict(layers=12, hidden=768, heads=12, ffn=3072, vocab=3000, pooling="cls", precision="mixed1"), 3, 160, 4),
])
def test_encoder_vs_oracle(kw, B, S, seed, lib_built):
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, synthetic_weights
    from oracle import bert_oracle
    cfg = EncoderConfig(**kw)
    w = synthetic_weights(cfg, seed)
    rng = np.random.default_rng(seed)
    ids = rng.integers(1000, cfg.vocab, size=(B, S)).astype(np.int32)
    lens = rng.integers(1, S + 1, size=B).astype(np.int32)
    lens[0], lens[-1] = S, 1                                        # full-length and single-token rows
    with Encoder(cfg, w) as enc:
        out = enc.encode(ids, lens)
        again = enc.encode(ids, lens)
    ref = bert_oracle.encode(w, cfg.as_dict(), ids, lens)
    assert np.isfinite(out).all()
    precise = cfg.precision in ("bf16x3", "mixed", "mixed1")
    assert (1.0 - _cos(out.astype(np.float64), ref)).max() <= ({"bf16x3": 1e-7, "mixed": 1e-6, "mixed1": 1e-5}[cfg.precision] if precise else TOL)
    if not cfg.normalize:
        np.testing.assert_allclose(np.linalg.norm(out, axis=1), np.linalg.norm(ref, axis=1), rtol=1e-4 if precise else 2e-2)
    np.testing.assert_array_equal(out, again)                      # deterministic


@pytest.mark.parametrize("hidden,ffn", [(384, 1536), (768, 3072)])
def test_attention_stage_and_block_boundaries(hidden, ffn, lib_built, monkeypatch):
    """attention_kernel streams keys in stages of 256 and blocks of 32, masks only the last block of a sequence, and
    lets the buffer bounds drop the rows past a sequence's end: lengths on every one of those edges, both head widths
    (d = 32 staged two heads at a time, d = 64), against the oracle; one sequence alone == the same sequence in the batch."""
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, synthetic_weights
    from oracle import bert_oracle
    _dbg(monkeypatch).set("small", "0")   # one sequence alone and in the batch through the same kernels: same bits
    cfg = EncoderConfig(layers=2, hidden=hidden, heads=12, ffn=ffn, vocab=3000)
    w = synthetic_weights(cfg, 31)
    rng = np.random.default_rng(31)
    lens = np.array([1, 7, 8, 31, 32, 33, 63, 64, 255, 256, 257, 287, 288, 289, 480, 511, 512], dtype=np.int32)
    ids = rng.integers(1000, cfg.vocab, size=(len(lens), 512)).astype(np.int32)
    with Encoder(cfg, w) as enc:
        out = enc.encode(ids, lens)
        alone = [enc.encode(ids[i:i + 1, :lens[i]], lens[i:i + 1])[0] for i in (0, 5, 10, 16)]
    ref = bert_oracle.encode(w, cfg.as_dict(), ids, lens)
    assert np.isfinite(out).all()
    assert (1.0 - _cos(out.astype(np.float64), ref)).max() <= TOL
    for j, i in enumerate((0, 5, 10, 16)):
        np.testing.assert_array_equal(out[i], alone[j])


@pytest.mark.parametrize("small", ["1", "0"])
def test_batch_composition_does_not_change_a_row(small, lib_built, monkeypatch):
    """Varlen packing: a sequence's embedding must not depend on its batch neighbours / padding ids.  Passes of the same
    kind return the same bits; a small pass (<= 2048 packed rows: encoder_small.hip sums the MLP's ffn chunks in another order)
    agrees with a large one to 1 - cos <= 1e-6, and bit for bit when MEMEX_HIP_DEBUG=small=0 routes it through the large-pass kernels."""
    _dbg(monkeypatch).set("small", small)
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, synthetic_weights
    cfg = EncoderConfig(layers=3, hidden=384, heads=12, ffn=1536, vocab=3000)
    w = synthetic_weights(cfg, 9)
    rng = np.random.default_rng(9)
    ids = rng.integers(1000, cfg.vocab, size=(7, 90)).astype(np.int32)
    lens = np.array([90, 13, 64, 1, 33, 90, 8], dtype=np.int32)
    with Encoder(cfg, w) as enc:
        full = enc.encode(ids, lens)
        ids2 = ids.copy()
        ids2[1, 13:] = 0
        alone = enc.encode(ids2[1:2, :13], lens[1:2])
        many = enc.encode(np.repeat(ids, 200, axis=0), np.repeat(lens, 200))     # 1400 seqs: several passes
    np.testing.assert_array_equal(full[1], alone[0])
    # hidden 384: the large-pass GEMMs (pgemm_kernel) and the small-pass ones sum k in the same order and share the
    # epilogue arithmetic, so the 1400-sequence call (two passes of >= 32768 rows) returns the 7-sequence call's bits
    np.testing.assert_array_equal(many.reshape(7, 200, -1)[:, 0], many.reshape(7, 200, -1)[:, 199])
    np.testing.assert_array_equal(many.reshape(7, 200, -1)[:, 5], many[::200])
    if small == "0":
        np.testing.assert_array_equal(many[::200], full)
    else:
        assert (1.0 - _cos(many[::200].astype(np.float64), full.astype(np.float64))).max() <= 1e-4
        assert np.abs(many[::200] - full).max() <= 1e-3


@pytest.mark.parametrize("layers,B,S,seed,ffn", [(6, 1, 16, 61, 1536), (12, 1, 128, 62, 1536), (6, 8, 32, 63, 1536), (3, 5, 77, 64, 1536),
                                                 (2, 1, 1, 65, 1536), (3, 2, 40, 66, 768), (3, 3, 33, 67, 384), (2, 1, 480, 68, 384),
                                                 (2, 4, 120, 69, 1152)])
def test_small_pass_matches_large_pass(layers, B, S, seed, ffn, lib_built, monkeypatch):
    """Query-time passes (<= 2048 packed rows, hidden 384) run encoder_small.hip -- one wave per 32 projection features, the MLP
    split over its ffn chunks -- with the operands, MFMA shape and rounding points of the large-pass kernels; only the f32
    summation order of the MLP's chunk products differs.  Both against the f64 oracle within the 1e-3 bar, and against each
    other far inside it."""
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, synthetic_weights
    from oracle import bert_oracle
    cfg = EncoderConfig(layers=layers, hidden=384, heads=12, ffn=ffn, vocab=3000)   # (ffn / 128 chunks: 3 .. 12 positions in the weight stream)
    w = synthetic_weights(cfg, seed)
    rng = np.random.default_rng(seed)
    ids = rng.integers(1000, cfg.vocab, size=(B, S)).astype(np.int32)
    lens = rng.integers(max(1, S // 2), S + 1, size=B).astype(np.int32)
    outs = []
    for small in ("1", "0"):
        _dbg(monkeypatch).set("small", small)
        with Encoder(cfg, w) as enc:
            outs.append(enc.encode(ids, lens))
            np.testing.assert_array_equal(outs[-1], enc.encode(ids, lens))
            if B > 1:                                                     # a row alone == the row in its batch (both small passes)
                np.testing.assert_array_equal(outs[-1][1], enc.encode(ids[1:2, :lens[1]], lens[1:2])[0])
    ref = bert_oracle.encode(w, cfg.as_dict(), ids, lens)
    for o in outs:
        assert np.isfinite(o).all()
        assert (1.0 - _cos(o.astype(np.float64), ref)).max() <= TOL
    assert (outs[0] != outs[1]).any() or layers * B * S <= 2, "the small-pass kernels did not run"
    # two bf16 forward passes whose f32 sums run in different orders round a few hidden-state elements differently per layer:
    # they sit as far from each other as each sits from the oracle (measured: 1e-7 at 6 layers x 16 tokens, 1e-5 at 12 x 128)
    d_small_large = (1.0 - _cos(outs[0].astype(np.float64), outs[1].astype(np.float64))).max()
    print(f"small vs large pass L{layers} B={B} S={S}: 1 - cos = {d_small_large:.2e}; vs oracle "
          f"{(1.0 - _cos(outs[0].astype(np.float64), ref)).max():.2e} / {(1.0 - _cos(outs[1].astype(np.float64), ref)).max():.2e}")
    assert d_small_large <= 1e-4


def test_host_call_paths_agree(lib_built):
    """mx_encoder_encode has two transports: query-sized calls (<= 4096 ids in <= 64 sequences) go through one pinned mapped
    page the kernels read and write in place, larger ones through copy commands; mx_encoder_encode_device takes device
    pointers.  Same kernels behind all three: the same bits."""
    import torch
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, synthetic_weights
    cfg = EncoderConfig(layers=2, hidden=384, heads=12, ffn=1536, vocab=3000)
    w = synthetic_weights(cfg, 81)
    rng = np.random.default_rng(81)
    with Encoder(cfg, w) as enc:
        for B, S in ((64, 64), (65, 63), (1, 4096 // 8), (3, 7)):
            ids = rng.integers(1000, cfg.vocab, size=(B, S)).astype(np.int32)
            lens = rng.integers(1, S + 1, size=B).astype(np.int32)
            host = enc.encode(ids, lens)
            d_out = torch.zeros((B, cfg.hidden), device="cuda")
            enc.encode_device(torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda(), d_out)
            np.testing.assert_array_equal(host, d_out.cpu().numpy())
            np.testing.assert_array_equal(host, enc.encode(ids, lens))


def test_bad_arguments(lib_built):
    from memex_amd import _lib
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, synthetic_weights
    cfg = EncoderConfig(layers=1, hidden=384, heads=12, ffn=1536, vocab=100)
    w = synthetic_weights(cfg, 0)
    with Encoder(cfg, w) as enc:
        with pytest.raises(_lib.MemexHipError):
            enc.encode(np.ones((1, 600), np.int32), np.array([600], np.int32))     # S > max_pos
        with pytest.raises(_lib.MemexHipError):
            enc.encode(np.ones((2, 8), np.int32), np.array([8, 0], np.int32))       # len < 1
        out = enc.encode(np.full((1, 4), 10**6, np.int32), np.array([4], np.int32))  # ids clamp, no fault
        assert np.isfinite(out).all()
    with pytest.raises(_lib.MemexHipError) as ei:
        Encoder(EncoderConfig(layers=1, hidden=64, heads=4, ffn=128, vocab=100), synthetic_weights(
            EncoderConfig(layers=1, hidden=64, heads=4, ffn=128, vocab=100), 0))
    assert ei.value.code == _lib.MX_EUNSUPPORTED


def test_embed_then_search_pipeline_matches_cpu_path(oracle, lib_built, tmp_path):
    """BASELINE config 1 shape (plumbing): segment -> embed -> add_vectors -> search, HIP vs CPU path.
    Identical f32 vectors => identical ids (bit-exact search); HIP-embedded vs oracle-embedded
    vectors => scores within 1e-3."""
    from memex_amd import embedding as E
    from memex_amd.storage import VectorData, get_vector_storage
    from memex_amd.weights import EncoderConfig, synthetic_weights
    from oracle import bert_oracle
    cfg = EncoderConfig(layers=2, hidden=384, heads=12, ffn=1536, vocab=30522, max_seq_length=128)
    w = synthetic_weights(cfg, 11)
    rng = np.random.default_rng(11)
    words = [f"tok{i}" for i in range(400)]
    doc = " ".join(rng.choice(words, size=3000))
    th, emb = E.SentenceEmbedder.spawn(E.ModelConfig(), weights=w, encoder_config=cfg, allow_synthetic=True)
    segs = emb.encode(doc)
    assert len(segs) == 1 + int(np.ceil((3000 - 256) / 170))
    qres = emb.encode_single("tok1 tok2 tok3 tok17")
    emb.shutdown()
    vecs = np.asarray([s.vector for s in segs], dtype=np.float32)
    # CPU path: oracle encoder on the same token ids
    tok = E.WhitespaceHashTokenizer(cfg.vocab)
    ids, lens = tok.encode_batch([s.content for s in segs], cfg.max_seq_length)
    ref = bert_oracle.encode(w, cfg.as_dict(), ids, lens).astype(np.float32)
    assert (1.0 - (vecs * ref).sum(1)).max() <= TOL
    vs = get_vector_storage(f"hnsw://{tmp_path}", "test")
    vs.add_vectors([VectorData(_id=f"seg-{i}", document_id="doc", text=s.content, vector=s.vector, segment_id=i)
                    for i, s in enumerate(segs)])
    q = np.asarray(qres.vector, dtype=np.float32)
    got = vs.search(q, 3)
    oi, od, os_, _ = oracle.search(vecs, q, 3)                     # same f32 vectors -> bit-exact
    assert [g[0] for g in got] == [f"seg-{int(i) - 1}" for i in oi[0]]
    np.testing.assert_array_equal(np.float32([g[1] for g in got]), os_[0])
    qi, ql = tok.encode_batch(["tok1 tok2 tok3 tok17"], cfg.max_seq_length)
    qref = bert_oracle.encode(w, cfg.as_dict(), qi, ql).astype(np.float32)[0]
    _, _, cpu_scores, _ = oracle.search(ref, qref, 3)              # all-CPU embed + search
    assert np.abs(np.float32([g[1] for g in got]) - cpu_scores[0]).max() <= TOL


def test_fused_layer_tail_equals_gemm_by_gemm_path(lib_built, monkeypatch):
    """tail_kernel (attention out-projection + Add&Norm + MLP + Add&Norm in one launch) rounds at the same
    points and accumulates in the same k order as the three GEMMs it replaces (MEMEX_HIP_DEBUG=unfused_tail=1):
    outputs must be bit-identical -- full passes, ragged lengths, a single short query and a non-default ffn
    width included."""
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, synthetic_weights
    for kw, B, S, seed in ((dict(layers=2, hidden=384, heads=12, ffn=1536, vocab=3000), 96, 512, 11),
                           (dict(layers=2, hidden=384, heads=12, ffn=768, vocab=3000), 300, 160, 12),
                           (dict(layers=3, hidden=384, heads=12, ffn=384, vocab=3000), 1, 9, 13),
                           (dict(layers=2, hidden=384, heads=12, ffn=1536, vocab=3000), 5, 77, 14)):
        cfg = EncoderConfig(**kw)
        w = synthetic_weights(cfg, seed)
        rng = np.random.default_rng(seed)
        ids = rng.integers(0, cfg.vocab, (B, S)).astype(np.int32)
        lens = rng.integers(S // 2 + S // 4, S + 1, B).astype(np.int32)
        outs = []
        _dbg(monkeypatch).set("small", "0")   # (small passes have kernels of their own: test_small_pass_matches_large_pass)
        for unfused in ("1", "0"):
            _dbg(monkeypatch).set("unfused_tail", unfused)
            with Encoder(cfg, w) as enc:
                outs.append(enc.encode(ids, lens))
        np.testing.assert_array_equal(outs[0], outs[1])


def test_large_passes_run_their_gemms_on_pgemm_kernel(lib_built, monkeypatch):
    """Passes of >= 32768 packed rows run the projections and MLP GEMMs on pgemm_kernel (encoder_pgemm.hip: persistent
    256 x 256 tiles, two wave rows half a phase apart) instead of gemm_kernel (MEMEX_HIP_DEBUG=pgemm=0).  Same k order and
    epilogue arithmetic: where only the schedule changes (hidden 384, GEMM-by-GEMM tail) the embeddings are
    bit-identical; the hidden-768 layer also moves its two LayerNorms behind the GEMM (y rounded to bf16 first,
    ln_rows_kernel), which must stay far inside the 1e-3 bar -- against the other path and against the f64 oracle."""
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, synthetic_weights
    from oracle import bert_oracle
    cases = ((dict(layers=2, hidden=768, heads=12, ffn=3072, vocab=3000, pooling="cls"), 96, 512, 41, False),   # bge-base layers, 43k rows
             (dict(layers=2, hidden=768, heads=12, ffn=3072, vocab=3000), 300, 160, 42, False),                # ragged, mean pooling
             (dict(layers=2, hidden=384, heads=12, ffn=1536, vocab=3000), 96, 512, 43, True),                  # QK projection + MLP GEMMs
             (dict(layers=2, hidden=768, heads=12, ffn=3072, vocab=3000), 40, 256, 44, None))                  # 10k rows: below the threshold
    for kw, B, S, seed, identical in cases:
        cfg = EncoderConfig(**kw)
        w = synthetic_weights(cfg, seed)
        rng = np.random.default_rng(seed)
        ids = rng.integers(0, cfg.vocab, (B, S)).astype(np.int32)
        lens = rng.integers(S // 2 + S // 4, S + 1, B).astype(np.int32)
        if cfg.hidden == 384:
            _dbg(monkeypatch).set("unfused_tail", "1")
        outs = []
        for pg in ("0", "1"):
            _dbg(monkeypatch).set("pgemm", pg)
            with Encoder(cfg, w) as enc:
                outs.append(enc.encode(ids, lens))
                np.testing.assert_array_equal(outs[-1], enc.encode(ids, lens))        # deterministic
        _dbg(monkeypatch).unset("unfused_tail")
        _dbg(monkeypatch).unset("pgemm")
        assert np.isfinite(outs[1]).all(), kw
        if identical or identical is None:
            np.testing.assert_array_equal(outs[0], outs[1])
        else:
            assert (outs[0] != outs[1]).any(), "the large-pass path did not run"
            assert (1.0 - _cos(outs[0].astype(np.float64), outs[1].astype(np.float64))).max() <= 5e-5, kw
        sub = slice(0, min(B, 16))
        ref = bert_oracle.encode_many(w, cfg.as_dict(), ids[sub], lens[sub])
        assert (1.0 - _cos(outs[1][sub].astype(np.float64), ref)).max() <= TOL, kw


def test_a_row_across_the_pass_size_regimes(lib_built):
    """A text's embedding depends (in its last bits) on how many packed rows share its pass -- three kernel sets, stated in
    include/memex_hip.h: hidden 384: small passes (<= 2048 rows: encoder_small.hip) / everything else; hidden 768: passes
    below / from 32768 rows (gemm_kernel with the fused Add&LayerNorm / pgemm_kernel + ln_rows_kernel).  Same rounding
    points, other f32 summation orders (and one more bf16 rounding of the pre-LayerNorm sum at >= 32768 rows): the same rows
    on either side of each threshold must agree far inside the 1e-3 bar."""
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, synthetic_weights
    rng = np.random.default_rng(71)
    for kw, S, b_small, b_large in ((dict(layers=4, hidden=768, heads=12, ffn=3072, vocab=3000, pooling="cls"), 512, 63, 64),
                                    (dict(layers=6, hidden=384, heads=12, ffn=1536, vocab=3000), 64, 31, 32)):
        cfg = EncoderConfig(**kw)
        w = synthetic_weights(cfg, 71)
        ids = rng.integers(1000, cfg.vocab, size=(b_large, S)).astype(np.int32)
        lens = np.full(b_large, S, dtype=np.int32)
        with Encoder(cfg, w) as enc:
            below = enc.encode(ids[:b_small], lens[:b_small])     # 63 x 512 = 32256 rows (+32 < 32768) / 31 x 64 = 1984 (+32 <= 2048)
            above = enc.encode(ids, lens)                         # 64 x 512 = 32768 / 32 x 64 = 2048 (+32 > 2048)
        d = (1.0 - _cos(below.astype(np.float64), above[:b_small].astype(np.float64))).max()
        print(f"hidden {cfg.hidden}: the same {b_small} rows in a pass below / above the threshold: 1 - cos = {d:.2e}")
        assert (below != above[:b_small]).any(), "both calls took the same kernels: the thresholds moved?"
        assert d <= 5e-5, (kw, d)


@pytest.mark.parametrize("kw,B,S,seed", [
    (dict(layers=6, hidden=384, heads=12, ffn=1536, vocab=3000), 12, 256, 51),                    # all-MiniLM-L6-v2 shape: fused tail
    (dict(layers=12, hidden=768, heads=12, ffn=3072, vocab=3000, pooling="cls"), 6, 200, 52),     # bge-base shape: gemm_kernel path
    (dict(layers=4, hidden=768, heads=12, ffn=3072, vocab=3000, pooling="cls"), 80, 512, 53),     # bge-base layers, 41k rows: pgemm_kernel + ln_rows_kernel
    (dict(layers=4, hidden=384, heads=12, ffn=1536, vocab=3000), 96, 512, 54),                    # MiniLM layers, large pass
])
@pytest.mark.parametrize("precision", ["bf16", "bf16x3", "mixed", "mixed1"])
def test_checkpoint_like_weights_stay_within_tolerance(kw, B, S, seed, precision, lib_built):
    """Trained checkpoints carry what random weights do not: outlier hidden dimensions (LayerNorm gains ~20, biases
    +-30 on a handful of dimensions in every layer) and attention logits of +-60.  Those are the bf16 hazards --
    activation range through the residual stream, the pre-LayerNorm sums the large-pass path rounds to bf16, the
    attention fast path's un-shifted exp2.  Real weights cannot be loaded offline (the reference downloads them,
    embedding.rs:99-100), so `checkpoint_like_weights` injects those features: cosine against the f64 oracle must stay
    within the same 1e-3."""
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, checkpoint_like_weights
    from oracle import bert_oracle
    cfg = EncoderConfig(**kw, precision=precision)
    w = checkpoint_like_weights(cfg, seed)
    rng = np.random.default_rng(seed)
    ids = rng.integers(0, cfg.vocab, (B, S)).astype(np.int32)
    lens = rng.integers(S // 2, S + 1, B).astype(np.int32)
    lens[0] = S
    with Encoder(cfg, w) as enc:
        out = enc.encode(ids, lens)
    assert np.isfinite(out).all()
    sub = slice(0, min(B, 8))
    ref = bert_oracle.encode_many(w, cfg.as_dict(), ids[sub], lens[sub])
    cos = _cos(out[sub].astype(np.float64), ref)
    assert (1.0 - cos).max() <= {"bf16": TOL, "bf16x3": 1e-6, "mixed": 1e-5, "mixed1": 1e-4}[precision], (kw, cos)
    # what the search sees: the cosines BETWEEN embeddings.  These weights put a large common component into every
    # embedding (pairwise cosines ~0.96) and most of its energy into five dimensions of magnitude 20-60, where a bf16 step
    # is 0.125-0.25: the row-wise cosine above is the easy half.  Measured (round 4): the pairwise cosines move by up to
    # 1.3e-3 with mean pooling (256+ tokens average the rounding noise) and by ~1e-2 with CLS pooling (one token, twelve
    # layers; gemm_kernel and pgemm_kernel paths alike).  Round 5: a numpy emulation of every rounding point
    # (scripts/encoder_rounding_sim.py) shows that this is NOT the residual stream -- with the hidden state, the GEMM results
    # and the final output all kept in f32 the error stays at 1e-2; with the weights ALONE rounded to bf16 it is 6e-3: a
    # logit of +-60 carries a bf16 error of 0.1 and moves its softmax weight by 10 % -- and that every operand needs 13+
    # significant bits for 1e-3.  precision="bf16x3" (encoder_precise.hip: split operands, f32 hidden state, f32 attention)
    # gives them 16 and must hold north_star's 1e-3 on the scores, CLS and mean alike; the bf16 default keeps its measured
    # bounds, labelled as what they are.
    o = out[sub].astype(np.float64)
    o /= np.linalg.norm(o, axis=1, keepdims=True)
    r = ref / np.linalg.norm(ref, axis=1, keepdims=True)
    pair = np.abs(o @ o.T - r @ r.T).max()
    print(f"checkpoint-like weights {kw['layers']}x{kw['hidden']} B={B} S={S} {precision}: max(1 - cos) = {(1.0 - cos).max():.2e}, "
          f"max |pairwise cosine error| = {pair:.2e}")
    if precision in ("bf16x3", "mixed", "mixed1"):
        # north_star: cosine scores within 1e-3 -- with the margin the rounding simulator promised (profiles/r6_encoder_rounding_sim.txt:
        # bf16x3 <= 4.7e-5, the mixed mode's MLP on fp16 weights x fp16 hi + lo activations <= 3.6e-5; its MLP on ONE fp16 product
        # <= 3.3e-4: mixed1 is asserted against the bar itself with a factor of two)
        assert pair <= {"bf16x3": 1e-4, "mixed": 2.5e-4, "mixed1": 5e-4}[precision], (kw, pair)
    else:
        assert pair <= (2.5e-2 if cfg.pooling == "cls" else 2.5e-3), (kw, pair)   # bf16 operands: measured, not the bar


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7])
def test_precision_modes_over_weight_seeds(seed, lib_built):
    """The four cases above are four draws of `checkpoint_like_weights`.  Other seeds of the same generator are harsher on the bge-base
    shape (12 x 768, CLS): seeds 1 and 4 put the all-f32 evaluation at 1.5e-5 / 4.0e-5 from the f64 oracle on pairwise scores where
    seed 52 has 1.5e-6, and the numpy emulation (scripts/encoder_rounding_sim.py) then says 3.4e-4 for three bf16 products
    everywhere, 1.2-1.3e-3 for the mixed mode, 2e-2 for mixed1 and 3e-2 for bf16 (profiles/r6_precision_modes_over_seeds.txt).
    So: MX_PREC_BF16X3 -- the mode the loaders pick for CLS-pooled hidden-768 models -- must hold north_star's 1e-3 on the scores on
    every seed; the cheaper modes are measured and labelled, not promised."""
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, checkpoint_like_weights
    from oracle import bert_oracle
    kw = dict(layers=12, hidden=768, heads=12, ffn=3072, vocab=3000, pooling="cls")
    B, S = 6, 200
    base = EncoderConfig(**kw)
    w = checkpoint_like_weights(base, seed)
    rng = np.random.default_rng(seed)
    ids = rng.integers(0, base.vocab, (B, S)).astype(np.int32)
    lens = rng.integers(S // 2, S + 1, B).astype(np.int32)
    lens[0] = S
    ref = bert_oracle.encode_many(w, base.as_dict(), ids, lens)
    r = ref / np.linalg.norm(ref, axis=1, keepdims=True)
    pair, row = {}, {}
    for precision in ("bf16x3", "mixed", "mixed1", "bf16"):
        with Encoder(EncoderConfig(**kw, precision=precision), w) as enc:
            o = enc.encode(ids, lens).astype(np.float64)
        assert np.isfinite(o).all()
        o /= np.linalg.norm(o, axis=1, keepdims=True)
        pair[precision] = float(np.abs(o @ o.T - r @ r.T).max())
        row[precision] = float((1.0 - (o * r).sum(1)).max())
    print(f"checkpoint-like weights 12x768 CLS seed {seed}: pairwise score error " + "  ".join(f"{k} {v:.2e}" for k, v in pair.items())
          + " | row-wise 1 - cos " + "  ".join(f"{k} {v:.2e}" for k, v in row.items()))
    assert row["bf16"] <= TOL, row                          # the parity bar of the ingest mode: each embedding against the oracle's
    assert pair["bf16x3"] <= 1e-3, pair                     # the bar, for the mode the loaders default to on this shape
    assert pair["mixed"] <= 5e-3 and pair["mixed1"] <= 6e-2 and pair["bf16"] <= 2e-1, pair   # measured, not the bar


def test_attention_fast_path_and_its_fallback(lib_built, monkeypatch):
    """attention_kernel's fast path applies no softmax shift (P = exp2(score)) and only falls back to the
    running maximum when a row sum says an exp2 may have overflowed or a row underflowed.  (a) ordinary
    weights: fast path == running-maximum loop (MEMEX_HIP_DEBUG=attn_safe=1) up to bf16 rounding of P, both within
    tolerance of the oracle; (b) query / key projections scaled so that scores reach several hundred either
    side of 0: every row of the fast path overflows or underflows there, so finite outputs that agree with
    the running-maximum loop mean the fallback ran and is right."""
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, synthetic_weights
    from oracle import bert_oracle
    cfg = EncoderConfig(layers=2, hidden=384, heads=12, ffn=1536, vocab=3000)
    rng = np.random.default_rng(21)
    B, S = 6, 512  # (keys stream through LDS in stages of 256: one-stage and two-stage sequences, ragged tails in either)
    ids = rng.integers(0, cfg.vocab, (B, S)).astype(np.int32)
    lens = np.array([512, 200, 97, 257, 33, 480], dtype=np.int32)
    for scale in (1.0, 4.0, 24.0):  # scores of a few units / of +-60 (fast path, some rows near its limits) / of several hundred
        w = synthetic_weights(cfg, 21)
        for name in list(w):
            if name.endswith("attention.self.query.weight") or name.endswith("attention.self.key.weight"):
                w[name] = (w[name] * scale).astype(np.float32)
        outs = []
        for safe in ("0", "1"):
            _dbg(monkeypatch).set("attn_safe", safe)
            with Encoder(cfg, w) as enc:
                outs.append(enc.encode(ids, lens))
        for o in outs:
            assert np.isfinite(o).all(), scale
        if scale == 1.0:  # (with scores in the hundreds bf16 q / k decide the arg max: no oracle comparison there)
            ref = bert_oracle.encode(w, cfg.as_dict(), ids, lens)
            for o in outs:
                cos = (o * ref).sum(1) / np.linalg.norm(o, axis=1) / np.linalg.norm(ref, axis=1)
                assert (1.0 - cos).max() <= TOL, cos
        cos = (outs[0] * outs[1]).sum(1) / np.linalg.norm(outs[0], axis=1) / np.linalg.norm(outs[1], axis=1)
        assert (1.0 - cos).max() <= 1e-4, (scale, cos)
        # d = 32 can stage two adjacent heads together (MEMEX_HIP_DEBUG=attn_pair=1): the same arithmetic per head, the same bits --
        # except that a redo takes both heads of a pair through the running-maximum loop (scale 4)
        _dbg(monkeypatch).set("attn_pair", "1")
        for i, safe in enumerate(("0", "1")):
            _dbg(monkeypatch).set("attn_safe", safe)
            with Encoder(cfg, w) as enc:
                o = enc.encode(ids, lens)
            if scale == 4.0 and safe == "0":
                cos = (o * outs[i]).sum(1) / np.linalg.norm(o, axis=1) / np.linalg.norm(outs[i], axis=1)
                assert (1.0 - cos).max() <= 1e-4, (scale, cos)
            else:
                assert np.array_equal(o, outs[i]), (scale, safe)
        _dbg(monkeypatch).unset("attn_pair")


def test_embedder_batches_concurrent_requests(lib_built):
    """Concurrent encode / encode_single calls on one embedder are embedded together; every caller
    still receives exactly what a lone call returns (a row's embedding is batch-independent)."""
    import threading
    from memex_amd import embedding as E
    from memex_amd.weights import EncoderConfig, synthetic_weights
    cfg = EncoderConfig(layers=2, hidden=384, heads=12, ffn=1536, vocab=30522, max_seq_length=128)
    w = synthetic_weights(cfg, 13)
    rng = np.random.default_rng(13)
    words = [f"tok{i}" for i in range(300)]
    texts = [" ".join(rng.choice(words, size=int(n))) for n in rng.integers(3, 700, 40)]
    th, emb = E.SentenceEmbedder.spawn(E.ModelConfig(), weights=w, encoder_config=cfg, allow_synthetic=True)
    alone = [emb.encode(t) for t in texts]                          # one at a time
    got = [None] * len(texts)
    errs = []

    def worker(i):
        try:
            got[i] = emb.encode(texts[i]) if i % 3 else [emb.encode_single(texts[i])]
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(len(texts))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    single = [emb.encode_single(t) for t in texts]
    emb.shutdown()
    assert not errs, errs[:2]
    for i in range(len(texts)):
        want = alone[i] if i % 3 else [single[i]]
        assert [r.content for r in got[i]] == [r.content for r in want]
        # (a lone short request is a small pass, the same text inside a combined batch may ride a large one: same rounding
        # points, another f32 summation order of the MLP chunks -- equal to ~1e-7 in cosine, not in bits)
        np.testing.assert_allclose(np.float32([r.vector for r in got[i]]), np.float32([r.vector for r in want]), atol=1e-3, rtol=0)


def test_keyed_encoder_is_shared_and_refcounted(lib_built):
    """mx_encoder_open(key): embedders spawned per request / per task (handlers.rs:61-63, tasks.rs:17)
    attach to ONE resident encoder instead of re-uploading the checkpoint."""
    import time
    from memex_amd import _lib
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, synthetic_weights
    cfg = EncoderConfig(layers=6, hidden=384, heads=12, ffn=1536, vocab=30522)
    w = synthetic_weights(cfg, 31)
    rng = np.random.default_rng(31)
    ids = rng.integers(1000, cfg.vocab, size=(3, 40)).astype(np.int32)
    lens = np.array([40, 7, 23], dtype=np.int32)
    a = Encoder(cfg, w, key="all-MiniLM-L6-v2@0")
    want = a.encode(ids, lens)
    t0 = time.perf_counter()
    for _ in range(50):                                             # the per-request pattern
        b = Encoder(cfg, None, key="all-MiniLM-L6-v2@0")           # attach: no weights needed, O(1)
        got = b.encode(ids, lens)
        b.close()
    dt = time.perf_counter() - t0
    np.testing.assert_array_equal(got, want)
    assert dt < 1.0, f"50 attach+encode+close took {dt:.2f} s"
    a.close()                                                       # last reference: the weights leave HBM
    with pytest.raises(_lib.MemexHipError):
        Encoder(cfg, None, key="all-MiniLM-L6-v2@0")
    other = EncoderConfig(layers=2, hidden=384, heads=12, ffn=1536, vocab=30522)
    c = Encoder(cfg, w, key="k2")
    with pytest.raises(_lib.MemexHipError):
        Encoder(other, synthetic_weights(other, 1), key="k2")      # same key, different configuration
    c.close()


@pytest.mark.parametrize("kw,B,S", [(dict(layers=3, hidden=384, heads=12, ffn=1536, vocab=3000, precision="bf16x3"), 9, 300),
                                    (dict(layers=2, hidden=768, heads=12, ffn=3072, vocab=3000, pooling="cls", precision="bf16x3"), 5, 512),
                                    (dict(layers=2, hidden=384, heads=6, ffn=768, vocab=3000, precision="bf16x3"), 6, 77)])
def test_split_bf16_attention_matches_the_f32_mfma_attention(kw, B, S, lib_built, monkeypatch):
    """MX_PREC_BF16X3's attention runs its two products as three bf16 MFMAs each (hi hi + lo hi + hi lo, keys of a k-step in
    the slot order that makes a lane's P registers its B operand); MEMEX_HIP_DEBUG=attn_f32=1 keeps the f32-MFMA kernel it replaced.
    Ragged lengths, key blocks cut by the sequence end, head dims 32 and 64: the two agree to the dropped lo x lo terms."""
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, checkpoint_like_weights
    cfg = EncoderConfig(**kw)
    w = checkpoint_like_weights(cfg, 5)
    rng = np.random.default_rng(5)
    ids = rng.integers(1000, cfg.vocab, size=(B, S)).astype(np.int32)
    lens = rng.integers(1, S + 1, size=B).astype(np.int32)
    lens[0], lens[1] = S, 33
    outs = []
    for f32_attn in ("0", "1"):
        _dbg(monkeypatch).set("attn_f32", f32_attn)
        with Encoder(cfg, w) as enc:
            outs.append(enc.encode(ids, lens).astype(np.float64))
    d = (1.0 - _cos(outs[0], outs[1])).max()
    pair = np.abs(outs[0] @ outs[0].T - outs[1] @ outs[1].T).max() if cfg.normalize else 0.0
    print(f"{kw['hidden']}/{kw['heads']}: split-bf16 vs f32-MFMA attention: 1 - cos = {d:.2e}, pairwise {pair:.2e}")
    assert (outs[0] != outs[1]).any(), "both encoders ran the same attention kernel"
    assert d <= 1e-7 and pair <= 1e-5, (d, pair)


@pytest.mark.parametrize("kw,B,S,seed", [
    (dict(layers=2, hidden=768, heads=12, ffn=3072, vocab=3000, pooling="cls", precision="bf16x3"), 80, 512, 61),   # every GEMM on pgemm_kernel
    (dict(layers=2, hidden=384, heads=12, ffn=1536, vocab=3000, precision="bf16x3"), 96, 512, 62),                # W1 (N = 1536) only: 1152 and 384 are no multiples of 256
    (dict(layers=2, hidden=768, heads=12, ffn=3072, vocab=3000, pooling="cls", precision="mixed"), 80, 512, 63),   # the fp16 two-product GEMMs of the mixed mode
    (dict(layers=2, hidden=384, heads=12, ffn=1536, vocab=3000, precision="mixed"), 96, 512, 64),
    (dict(layers=2, hidden=768, heads=12, ffn=3072, vocab=3000, pooling="cls", precision="mixed1"), 80, 512, 65),  # ... and the one-product ones
    (dict(layers=2, hidden=384, heads=12, ffn=1536, vocab=3000, precision="mixed1"), 96, 512, 66)])
def test_split_operand_mode_on_pgemm_kernel(kw, B, S, seed, lib_built, monkeypatch):
    """MX_PREC_BF16X3 in large passes: its GEMMs run on pgemm_kernel where the shape allows (EPI_F32 / EPI_GELU_SPLIT through the
    wave-private scratch tile).  Same k order, same products, f32 sums in the same order as gemm_kernel: bit-identical
    embeddings with MEMEX_HIP_DEBUG=pgemm=0, and within the mode's bars of the f64 oracle."""
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, checkpoint_like_weights
    from oracle import bert_oracle
    cfg = EncoderConfig(**kw)
    w = checkpoint_like_weights(cfg, seed)
    rng = np.random.default_rng(seed)
    ids = rng.integers(1000, cfg.vocab, (B, S)).astype(np.int32)
    lens = rng.integers(S // 2, S + 1, B).astype(np.int32)
    outs = []
    for pg in ("0", "1"):
        _dbg(monkeypatch).set("pgemm", pg)
        with Encoder(cfg, w) as enc:
            outs.append(enc.encode(ids, lens))
    print(f"hidden {cfg.hidden}: pgemm vs gemm max |diff| {np.abs(outs[0] - outs[1]).max():.3e}, rows differing {(outs[0] != outs[1]).any(axis=1).sum()} of {B}")
    sub = np.arange(0, B, 9)
    ref = bert_oracle.encode(w, cfg.as_dict(), ids[sub], lens[sub])
    print(f"gemm_kernel path vs oracle: {(1.0 - _cos(outs[0][sub].astype(np.float64), ref)).max():.2e}")
    d = (1.0 - _cos(outs[1][sub].astype(np.float64), ref)).max()
    pair = np.abs(outs[1][sub].astype(np.float64) @ outs[1][sub].astype(np.float64).T - ref @ ref.T).max()
    print(f"bf16x3 on pgemm_kernel, hidden {cfg.hidden}: 1 - cos = {d:.2e}, pairwise {pair:.2e}")
    assert d <= {"bf16x3": 1e-6, "mixed": 1e-5, "mixed1": 1e-4}[cfg.precision] and pair <= 1e-3, (d, pair)
    np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.parametrize("kw,B,S,seed", [
    (dict(layers=12, hidden=768, heads=12, ffn=3072, vocab=3000, pooling="cls"), 1, 16, 81),          # bge-base: one short query
    (dict(layers=6, hidden=768, heads=12, ffn=3072, vocab=3000, max_pos=514, type_vocab=1, ln_eps=1e-5, pos_offset=2), 5, 90, 82),   # all-distilroberta-v1
    (dict(layers=3, hidden=768, heads=12, ffn=3072, vocab=3000), 14, 128, 83),                         # 1792 rows: just below the threshold
    (dict(layers=2, hidden=768, heads=12, ffn=1536, vocab=3000), 3, 200, 84)])                         # another ffn width (4 k-chunks)
def test_hidden_768_small_passes_split_k(kw, B, S, seed, lib_built, monkeypatch):
    """Passes of <= 2048 rows of a hidden-768 model run their two Add & LayerNorm GEMMs split over k (f32 partials from
    gemm_kernel<EPI_F32>, reduce_res_ln_kernel behind them) instead of one 64 x 768 workgroup per row tile looping over all of k
    (80 us per layer for one query).  Same rounding points, another f32 summation order: within 1e-5 of the fused form
    (MEMEX_HIP_DEBUG=splitk=0), and within the usual bar of the f64 oracle."""
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, checkpoint_like_weights
    from oracle import bert_oracle
    cfg = EncoderConfig(**kw)
    w = checkpoint_like_weights(cfg, seed)
    rng = np.random.default_rng(seed)
    ids = rng.integers(1000, cfg.vocab, size=(B, S)).astype(np.int32)
    lens = rng.integers(1, S + 1, size=B).astype(np.int32)
    lens[0] = S
    outs = []
    for sk in ("1", "0"):
        _dbg(monkeypatch).set("splitk", sk)
        with Encoder(cfg, w) as enc:
            outs.append(enc.encode(ids, lens))
            np.testing.assert_array_equal(outs[-1], enc.encode(ids, lens))
    ref = bert_oracle.encode(w, cfg.as_dict(), ids, lens)
    d_ref = (1.0 - _cos(outs[0].astype(np.float64), ref)).max()
    d_pair = (1.0 - _cos(outs[0].astype(np.float64), outs[1].astype(np.float64))).max()
    print(f"split-k small pass: vs fused {d_pair:.2e}, vs oracle {d_ref:.2e} (fused vs oracle {(1.0 - _cos(outs[1].astype(np.float64), ref)).max():.2e})")
    assert (outs[0] != outs[1]).any(), "both encoders ran the same kernels"
    assert d_ref <= TOL and d_pair <= 1e-4, (d_ref, d_pair)


@pytest.mark.gpu
@pytest.mark.parametrize("kw,seed", [
    (dict(layers=2, hidden=384, heads=12, ffn=1536, vocab=3000), 81),                       # tail_kernel / gemm_kernel / pgemm QK
    (dict(layers=2, hidden=768, heads=12, ffn=3072, vocab=3000, pooling="cls"), 82),        # pgemm_kernel + ln_rows_kernel
    (dict(layers=1, hidden=384, heads=12, ffn=1536, vocab=3000, precision="bf16x3"), 83),   # the split-operand layer
])
def test_full_passes_compute_whole_tiles_only(kw, seed, lib_built):
    """The GEMMs, the layer tail and the LayerNorm passes run over round_up(rows, 256) rows, the workspace (and the rows a key
    block reads past the last sequence) over round_up(rows + 32, 256): a pass whose packed rows fill their last 256-row tile --
    the ingest shape, 256 chunks x 512 tokens -- no longer multiplies a 513th tile of padding.  Rows between the two are then
    never written; run such passes on a workspace a LARGER pass has left dirty and hold every sequence against the oracle,
    the last ones (whose key blocks reach into the unwritten rows) in particular."""
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, synthetic_weights
    from oracle import bert_oracle
    cfg = EncoderConfig(**kw)
    w = synthetic_weights(cfg, seed)
    rng = np.random.default_rng(seed)
    S = 512
    precise = cfg.precision == "bf16x3"
    tol = 1e-7 if precise else TOL
    with Encoder(cfg, w) as enc:
        big_ids = rng.integers(1000, cfg.vocab, size=(80, S)).astype(np.int32)
        enc.encode(big_ids, np.full(80, S, dtype=np.int32))                       # 40960 rows: dirties the workspace
        for lens in (np.full(64, S), np.full(8, S),                                # 32768 / 4096 rows: a multiple of 256 exactly
                     np.r_[np.full(63, S), [S - 16]],                              # 32752: 240 of the last tile (rows + 32 spills over)
                     np.r_[np.full(3, S), [505]]):                                 # 2048 exactly through a length that is not 8-aligned
            lens = lens.astype(np.int32)
            B = len(lens)
            ids = rng.integers(1000, cfg.vocab, size=(B, S)).astype(np.int32)
            out = enc.encode(ids, lens)
            assert np.isfinite(out).all()
            sub = np.r_[0:1, max(1, B - 3):B]                      # the first sequence and the last three
            ref = bert_oracle.encode_many(w, cfg.as_dict(), ids[sub], lens[sub])
            assert (1.0 - _cos(out[sub].astype(np.float64), ref)).max() <= tol, (kw, lens[-3:])
            np.testing.assert_array_equal(out, enc.encode(ids, lens))


@pytest.mark.gpu
def test_short_sequence_passes_pair_heads(lib_built, monkeypatch):
    """Head dim 32, passes of short sequences: the attention's per-item fixed cost, not exp2, bounds them.
    attention_short_kernel (a plain grid, one workgroup of up to four waves per item, K / V^T shared through LDS behind one
    barrier) takes the passes whose longest sequence has <= 128 tokens; full passes (>= 1024 items) up to 256 tokens stage two
    heads per item of the staged kernel.  Both
    forms repeat attention_kernel's arithmetic instruction for instruction: whatever the automatic rule picks must equal the
    one-head staged kernel (MEMEX_HIP_DEBUG=attn_short=0, MEMEX_HIP_DEBUG=attn_pair=0) bit for bit, and so must each form when forced --
    under ordinary weights, on the running-maximum path (MEMEX_HIP_DEBUG=attn_safe=1), and with scores of several hundred, where
    every row leaves the fast path and is redone."""
    from memex_amd.encoder import Encoder
    from memex_amd.weights import EncoderConfig, synthetic_weights
    from oracle import bert_oracle
    cfg = EncoderConfig(layers=3, hidden=384, heads=12, ffn=1536, vocab=3000)
    rng = np.random.default_rng(91)

    def encode(w, ids, lens, **env):
        for kname in ("attn_short", "attn_short_lds", "attn_pair", "attn_safe"):
            _dbg(monkeypatch).unset(kname)
        for kname, v in env.items():
            _dbg(monkeypatch).set(kname, v)
        with Encoder(cfg, w) as enc:
            out = enc.encode(ids, lens)
        for kname in env:
            _dbg(monkeypatch).unset(kname)
        return out

    for scale in (1.0, 24.0):
        w = synthetic_weights(cfg, 91)
        for name in list(w):
            if scale != 1.0 and (name.endswith("attention.self.query.weight") or name.endswith("attention.self.key.weight")):
                w[name] = (w[name] * scale).astype(np.float32)
        # (B, S, shortest length); the last two: too few items for the automatic rule, forced forms only
        for B, S, lo in ((128, 128, 1), (100, 97, 30), (96, 256, 200), (90, 300, 40), (40, 64, 1), (3, 128, 100)):
            ids = rng.integers(1000, cfg.vocab, size=(B, S)).astype(np.int32)
            lens = rng.integers(lo, S + 1, size=B).astype(np.int32)
            lens[0] = S
            base = encode(w, ids, lens, attn_short="0", attn_pair="0")
            assert np.isfinite(base).all()
            auto = encode(w, ids, lens)
            if scale == 1.0:
                np.testing.assert_array_equal(auto, base, err_msg=f"automatic choice, {B} x {S}")
            else:  # (the automatic choice may be the head pairs: a redo takes the pair partner along, last-bit differences)
                assert (1.0 - _cos(auto.astype(np.float64), base.astype(np.float64))).max() <= 1e-4, (B, S, scale)
            np.testing.assert_array_equal(encode(w, ids, lens, attn_short="1"), base, err_msg=f"short kernel, {B} x {S}, scale {scale}")
            np.testing.assert_array_equal(encode(w, ids, lens, attn_short="1", attn_short_lds="0"), base,
                                          err_msg=f"short kernel, fragments from global memory, {B} x {S}, scale {scale}")
            if scale == 1.0:  # (a redo takes a pair partner along through the running-maximum loop: last-bit differences, see the header)
                np.testing.assert_array_equal(encode(w, ids, lens, attn_short="0", attn_pair="1"), base,
                                              err_msg=f"head pairs, {B} x {S}")
            safe = encode(w, ids, lens, attn_short="0", attn_pair="0", attn_safe="1")
            np.testing.assert_array_equal(encode(w, ids, lens, attn_short="1", attn_safe="1"), safe,
                                          err_msg=f"short kernel, running maximum, {B} x {S}, scale {scale}")
            if scale == 1.0:
                sub = np.r_[0:min(3, B), max(3, B - 3):B]
                ref = bert_oracle.encode_many(w, cfg.as_dict(), ids[sub], lens[sub])
                assert (1.0 - _cos(base[sub].astype(np.float64), ref)).max() <= TOL, (B, S)
