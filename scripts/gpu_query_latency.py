"""Query-time embedding + search latency probe (not a test)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from memex_amd.encoder import Encoder
from memex_amd import weights as W
for cfg, name in ((W.ALL_MINILM_L12_V2, "L12"), (W.ALL_MINILM_L6_V2, "L6")):
    enc = Encoder(cfg, W.synthetic_weights(cfg, 0))
    for B, S in ((1, 16), (1, 128), (8, 32), (55, 256)):
        ids = np.random.default_rng(0).integers(1000, cfg.vocab, (B, S)).astype(np.int32)
        lens = np.full((B,), S, dtype=np.int32)
        for _ in range(3): enc.encode(ids, lens)
        t0 = time.perf_counter()
        n = 50
        for _ in range(n): enc.encode(ids, lens)
        dt = (time.perf_counter() - t0) / n
        print(f"{name} B={B} S={S}: {dt*1e3:.3f} ms per encode call (host API)", flush=True)
    enc.close()
