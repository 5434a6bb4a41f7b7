"""Fused layer tail vs the GEMM-by-GEMM path on the same inputs: bit-equality and, if not, how far apart."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memex_amd.encoder import Encoder
from memex_amd.weights import EncoderConfig, synthetic_weights
cfg = EncoderConfig(layers=int(os.environ.get("LAYERS", "1")), hidden=384, heads=12, ffn=1536, vocab=3000)
w = synthetic_weights(cfg, 11)
rng = np.random.default_rng(11)
B, S = 96, 512
ids = rng.integers(0, cfg.vocab, (B, S)).astype(np.int32)
lens = rng.integers(S // 2 + S // 4, S + 1, B).astype(np.int32)
outs = []
for unfused in ("1", "0"):
    os.environ["MEMEX_HIP_UNFUSED_MLP"] = unfused
    with Encoder(cfg, w) as enc:
        outs.append(enc.encode(ids, lens))
a, b = outs
print("equal:", np.array_equal(a, b), "max abs diff", np.abs(a - b).max(), "rows differing", int((a != b).any(1).sum()), "of", B)
cos = (a * b).sum(1) / np.linalg.norm(a, axis=1) / np.linalg.norm(b, axis=1)
print("min cos", cos.min())
