"""GPU bring-up diagnostics for the encoder vs oracle/bert_oracle.py (not a test)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from memex_amd.encoder import Encoder
from memex_amd.weights import EncoderConfig, synthetic_weights, pack_weights
from oracle import bert_oracle

def case(cfg, B, S, seed, ragged=True):
    rng = np.random.default_rng(seed)
    w = synthetic_weights(cfg, seed)
    ids = rng.integers(1000, cfg.vocab, size=(B, S)).astype(np.int32)
    lens = rng.integers(max(1, S // 4), S + 1, size=B).astype(np.int32) if ragged else np.full(B, S, np.int32)
    lens[0] = S
    if B > 1: lens[1] = 1
    t0 = time.time()
    with Encoder(cfg, w) as enc:
        t1 = time.time()
        out = enc.encode(ids, lens)
        t2 = time.time()
    ref = bert_oracle.encode(w, cfg.as_dict(), ids, lens)
    cos = (out.astype(np.float64) * ref).sum(1) / np.linalg.norm(out, axis=1) / np.linalg.norm(ref, axis=1)
    err = np.abs(out - ref).max()
    print(f"L={cfg.layers} H={cfg.hidden} F={cfg.ffn} pool={cfg.pooling} B={B} S={S}: min cos={cos.min():.6f} max|d|={err:.2e} "
          f"finite={np.isfinite(out).all()} norm={np.linalg.norm(out,axis=1)[:3]} create={t1-t0:.2f}s enc={t2-t1:.3f}s")
    return cos.min()

small_vocab = 2000
case(EncoderConfig(layers=1, hidden=384, heads=12, ffn=1536, vocab=small_vocab), 4, 32, 1)
case(EncoderConfig(layers=2, hidden=384, heads=12, ffn=1536, vocab=small_vocab), 5, 128, 2)
case(EncoderConfig(layers=6, hidden=384, heads=12, ffn=1536, vocab=small_vocab), 4, 256, 3)
case(EncoderConfig(layers=6, hidden=384, heads=12, ffn=1536, vocab=small_vocab), 3, 512, 4)
case(EncoderConfig(layers=2, hidden=768, heads=12, ffn=3072, vocab=small_vocab, pooling="cls"), 4, 128, 5)
case(EncoderConfig(layers=12, hidden=768, heads=12, ffn=3072, vocab=small_vocab, pooling="cls"), 3, 200, 6)
