import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_centred_gpu import cone_rows
from oracle.search_oracle import COracle
from memex_amd.index import FlatIndex
oracle = COracle()
rng = np.random.default_rng(12)
d, n = 384, 30000
X = cone_rows(rng, n, d)
X[5] *= np.float32(1e-19); X[6] *= np.float32(1e18); X[7] *= np.float32(3e-16); X[50] = 0.0; X[51] = 0.0
Q = cone_rows(rng, 40, d)
Q[1] = rng.standard_normal(d).astype(np.float32); Q[2] = -Q[3]; Q[4] = Q[4] * np.float32(1e-30); Q[5] = Q[5] * np.float32(1e30); Q[6] = X[6]; Q[7] = X[5]
for kind in ("bf16", "i8", False):
    with FlatIndex(d) as idx:
        idx.add(X)
        idx.set_filter_copy(kind)
        st = idx.stats()
        for k in (1, 10, 64):
            ids, sc, di, nf = idx.search(Q, k)
            oi, od, os_, onf = oracle.search(X, Q, k)
            bad = [q for q in range(len(Q)) if not (np.array_equal(ids[q], oi[q]) and np.array_equal(di[q].view(np.uint32), od[q].view(np.uint32)))]
            print(f"kind {kind} centred {st.filter_centred} k {k}: mismatching queries {bad}  fallbacks {idx.stats().fallback_queries}")
            for q in bad[:2]:
                print("  q", q, "got ids", ids[q][:12].tolist(), "dist", di[q][:6].tolist(), "nf", nf[q])
                print("  q", q, "want   ", oi[q][:12].tolist(), "dist", od[q][:6].tolist(), "nf", onf[q])
