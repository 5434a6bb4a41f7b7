"""Encoder oracle (oracle/bert_oracle.py) vs the committed golden vectors (no GPU).

PARITY UNPINNED by the reference: no reference test asserts an embedding value
(lib/libmemex/src/llm/embedding.rs:204-217 checks a token count and needs the network).  The golden
outputs were produced by transformers.BertModel in f64 on the same seeded weights
(tests/golden/make_encoder_golden.py) -- an independent implementation of the published architecture.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))


def golden_cases():
    import make_encoder_golden as mg
    from memex_amd.weights import EncoderConfig, synthetic_weights
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_golden.npz"))
    for name, (kw, B, S, seed) in mg.CASES.items():
        cfg = EncoderConfig(**kw)
        ids, lens = mg.inputs(cfg, B, S, seed)
        yield name, cfg, synthetic_weights(cfg, seed), ids, lens, g[name + "_out"]


def test_oracle_matches_transformers_golden():
    from oracle import bert_oracle
    for name, cfg, w, ids, lens, gold in golden_cases():
        out = bert_oracle.encode(w, cfg.as_dict(), ids, lens)
        assert np.abs(out - gold).max() < 1e-9, name
        np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-12)


def test_padding_is_ignored_and_mask_semantics():
    from memex_amd.weights import EncoderConfig, synthetic_weights
    from oracle import bert_oracle
    cfg = EncoderConfig(layers=1, hidden=64, heads=4, ffn=128, vocab=300)
    w = synthetic_weights(cfg, 3)
    rng = np.random.default_rng(3)
    ids = rng.integers(5, 300, size=(2, 20)).astype(np.int32)
    lens = np.array([20, 7], dtype=np.int32)
    a = bert_oracle.encode(w, cfg.as_dict(), ids, lens)
    ids2 = ids.copy()
    ids2[1, 7:] = 1                                   # different padding ids must not matter
    b = bert_oracle.encode(w, cfg.as_dict(), ids2, lens)
    assert np.abs(a - b).max() < 1e-6                 # -10000 additive mask: exp(-1e4) == 0 in f64 too
    c = bert_oracle.encode(w, cfg.as_dict(), ids[1:2, :7], np.array([7]))
    assert np.abs(a[1] - c[0]).max() < 1e-6


def test_weight_packing_layout(lib_built):
    from memex_amd.encoder import Encoder
    from memex_amd.weights import ALL_MINILM_L6_V2, BGE_BASE_EN, EncoderConfig, pack_weights, synthetic_weights, tensor_order
    assert Encoder.weight_bytes(ALL_MINILM_L6_V2) == 22_565_376 * 4            # 22,565,376 params
    assert Encoder.weight_bytes(BGE_BASE_EN) == sum(int(np.prod(s)) for _, s in tensor_order(BGE_BASE_EN)) * 4
    cfg = EncoderConfig(layers=1, hidden=384, heads=12, ffn=1536, vocab=50)
    w = synthetic_weights(cfg, 0)
    blob = pack_weights({"bert." + k: v for k, v in w.items()}, cfg)           # prefixed names accepted
    assert blob.nbytes == Encoder.weight_bytes(cfg)
    off = 50 * 384 + 512 * 384 + 2 * 384
    np.testing.assert_array_equal(blob[off:off + 384], w["embeddings.LayerNorm.weight"])
    with pytest.raises(KeyError):
        pack_weights({}, cfg)


def test_segment_text_windows_and_model_gate():
    from memex_amd import embedding as E
    text = " ".join(f"w{i}" for i in range(600))
    segs = E.segment_text(E.ModelConfig(), text)
    # 256-token windows advancing by 256-86 = 170 (reference embedding.rs:64-73,173-177)
    starts = [int(s.split()[0][1:]) for s in segs]
    assert starts == [0, 170, 340, 510] and len(segs[0].split()) == 256 and segs[-1].split()[-1] == "w599"
    assert E.segment_text(E.ModelConfig(), "short text") == ["short text"]
    with pytest.raises(E.SetupError):
        E.segment_text(E.ModelConfig(model=E.EmbeddingsModelType.SentenceT5Base), "x")   # embedding.rs:160
    d = E.ModelConfig()
    assert (d.model, d.max_length, d.stride) == (E.EmbeddingsModelType.AllMiniLmL12V2, 256, 86)


