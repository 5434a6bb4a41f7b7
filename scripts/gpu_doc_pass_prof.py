"""One document's encoder pass (the reference's model.encode batch: ~72 windows x <= 128 tokens, all-MiniLM-L12-v2 shape) under
rocprofv3 --kernel-trace --stats: which kernels a mid-size pass spends its time in.  usage: gpu_doc_pass_prof.py [windows] [tokens]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from memex_amd.encoder import Encoder
from memex_amd import weights as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 72
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cfg = W.ALL_MINILM_L12_V2
enc = Encoder(cfg, W.synthetic_weights(cfg, 0))
rng = np.random.default_rng(0)
ids = rng.integers(1000, cfg.vocab, (B, S)).astype(np.int32)
lens = np.full((B,), S, dtype=np.int32)
import time
for _ in range(5): enc.encode(ids, lens)
t0 = time.perf_counter()
for _ in range(50): enc.encode(ids, lens)
print(f"{B} x {S}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per encode call (host API)")
