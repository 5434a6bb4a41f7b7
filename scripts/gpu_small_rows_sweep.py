"""Where the small-pass layer should hand over to the bulk kernels: per-call time of mx_encoder_encode for B windows of S tokens
under the threshold given by MEMEX_HIP_DEBUG=small_rows=N (read at encoder creation).  usage: gpu_small_rows_sweep.py [l6|l12] [S]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from memex_amd.encoder import Encoder
from memex_amd import weights as W
cfg = W.ALL_MINILM_L12_V2 if len(sys.argv) < 2 or sys.argv[1] == "l12" else W.ALL_MINILM_L6_V2
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
enc = Encoder(cfg, W.synthetic_weights(cfg, 0))
rng = np.random.default_rng(0)
out = []
for B in (2, 4, 8, 16, 24, 32, 48, 64, 96, 128):
    ids = rng.integers(1000, cfg.vocab, (B, S)).astype(np.int32)
    lens = np.full((B,), S, dtype=np.int32)
    for _ in range(5): enc.encode(ids, lens)
    t0 = time.perf_counter()
    for _ in range(40): enc.encode(ids, lens)
    out.append(f"{B * S}:{(time.perf_counter() - t0) / 40 * 1e3:.3f}")
print(f"DEBUG={os.environ.get('MEMEX_HIP_DEBUG', 'default')} S={S} rows:ms  " + "  ".join(out))
