"""Score errors of the precision modes against MX_PREC_BF16X3 on bench.py's sample (checkpoint_like_weights seed 52, 8 x 100..200 tokens)
and on a few more seeds: max |cos(e_i, e_j) - cos_x3(e_i, e_j)| per mode.  usage: gpu_precise_score_errors.py [seeds]"""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from memex_amd.encoder import Encoder
from memex_amd import weights as W

seeds = [int(x) for x in sys.argv[1:]] or [52, 1, 2, 3, 4, 5]
for name, base in (("all-MiniLM-L6-v2", W.ALL_MINILM_L6_V2), ("bge-base-en", W.BGE_BASE_EN)):
    worst = {}
    for seed in seeds:
        small = dataclasses.replace(base, layers=min(base.layers, 12), vocab=3000)
        wck = W.checkpoint_like_weights(small, seed)
        rng = np.random.default_rng(seed)
        cid = rng.integers(0, small.vocab, (8, 200)).astype(np.int32)
        cln = rng.integers(100, 201, 8).astype(np.int32)
        embs = {}
        for prec in ("bf16", "mixed1", "mixed", "bf16x3"):
            c2 = dataclasses.replace(small, precision=prec)
            with Encoder(c2, W.pack_weights(wck, c2)) as e2:
                v = e2.encode(cid, cln).astype(np.float64)
            embs[prec] = v / np.linalg.norm(v, axis=1, keepdims=True)
        ref = embs["bf16x3"] @ embs["bf16x3"].T
        err = {p: float(np.abs(embs[p] @ embs[p].T - ref).max()) for p in ("bf16", "mixed1", "mixed")}
        for p, v in err.items(): worst[p] = max(worst.get(p, 0.0), v)
        print(f"{name} seed {seed}: " + "  ".join(f"{p} {v:.2e}" for p, v in err.items()), flush=True)
    print(f"{name} worst of {len(seeds)} seeds: " + "  ".join(f"{p} {v:.2e}" for p, v in worst.items()), flush=True)
