"""Query-time embedding latency probe (not a test): per-call wall time of mx_encoder_encode (host API), median / p99 / max.
usage: gpu_query_latency.py [models: l12,l6,bge,roberta] [shapes: 1x16,1x128,8x32,55x256]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from memex_amd.encoder import Encoder
from memex_amd import weights as W
models = (sys.argv[1] if len(sys.argv) > 1 else "l12,l6").split(",")
shapes = [tuple(int(v) for v in s.split("x")) for s in (sys.argv[2] if len(sys.argv) > 2 else "1x16,1x128,8x32,55x256").split(",")]
for name in models:
    cfg = {"l12": W.ALL_MINILM_L12_V2, "l6": W.ALL_MINILM_L6_V2, "bge": W.BGE_BASE_EN, "roberta": W.ALL_DISTILROBERTA_V1}[name]
    enc = Encoder(cfg, W.synthetic_weights(cfg, 0))
    for B, S in shapes:
        ids = np.random.default_rng(0).integers(1000, cfg.vocab, (B, S)).astype(np.int32)
        lens = np.full((B,), S, dtype=np.int32)
        for _ in range(5): enc.encode(ids, lens)
        ts = []
        for _ in range(200):
            t0 = time.perf_counter()
            enc.encode(ids, lens)
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = np.asarray(ts)
        print(f"{name} B={B} S={S}: mean {ts.mean():.3f} ms, p50 {np.percentile(ts, 50):.3f}, p99 {np.percentile(ts, 99):.3f}, max {ts.max():.3f} "
              f"per encode call (host API)", flush=True)
    enc.close()
