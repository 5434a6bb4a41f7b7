"""GPU bring-up diagnostics for the flat index (not a test; prints details).  Run via gpurun."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from memex_amd.index import FlatIndex
from memex_amd import _lib
from oracle.search_oracle import COracle

orc = COracle()

def case(n, d, B, k, seed=0, mode=_lib.MX_SEARCH_AUTO, special=None):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d), dtype=np.float32)
    Q = rng.standard_normal((B, d), dtype=np.float32)
    if special == "dups":
        X[n // 2: n // 2 + 40] = X[7]
        Q[0] = X[7] * 3.0
    if special == "zero":
        X[5] = 0; X[n - 1] = 0; Q[1] = 0
    idx = FlatIndex(d)
    idx.set_profiling(True)
    idx.set_search_mode(mode)
    idx.add(X)
    t0 = time.time()
    ids, sc, di, nf = idx.search(Q, k)
    t1 = time.time()
    oi, od, os_, onf = orc.search(X, Q, k)
    st = idx.stats()
    ok_ids = np.array_equal(ids, oi)
    ok_d = np.array_equal(di.view(np.uint32), od.view(np.uint32))
    ok_s = np.array_equal(sc.view(np.uint32), os_.view(np.uint32))
    print(f"n={n} d={d} B={B} k={k} mode={mode} sp={special}: ids={ok_ids} dists={ok_d} scores={ok_s} nf={np.array_equal(nf,onf)} "
          f"fallback={st.fallback_queries} cand={st.candidates} max_err={st.max_abs_err:.2e} scan_ms={st.scan_ms:.3f} t={t1-t0:.3f}s")
    if not ok_ids:
        bad = np.argwhere(ids != oi)
        print("   first mismatches:", bad[:5].tolist(), ids[bad[0][0]][:k], oi[bad[0][0]][:k], di[bad[0][0]][:k], od[bad[0][0]][:k])
    idx.close()
    return ok_ids and ok_d and ok_s

allok = True
# reference KAT (local.rs:175-213)
idx = FlatIndex(3)
idx.add(np.array([[0.0, 0.1, 0.2], [0.1, 0.1, 0.1], [0.3, 0.2, 0.1]], dtype=np.float32))
ids, sc, di, nf = idx.search(np.array([0.1, 0.1, 0.1], dtype=np.float32), 3)
print("KAT", ids, sc, di, nf)
idx.close()
for args in [dict(n=3, d=3, B=1, k=3), dict(n=100, d=3, B=4, k=10), dict(n=1000, d=384, B=16, k=10),
             dict(n=1000, d=384, B=16, k=10, mode=_lib.MX_SEARCH_EXACT),
             dict(n=5000, d=100, B=33, k=7), dict(n=20000, d=768, B=256, k=10),
             dict(n=100000, d=384, B=256, k=10), dict(n=100000, d=384, B=300, k=10, seed=3),
             dict(n=50000, d=384, B=8, k=10, special="dups"), dict(n=50000, d=384, B=8, k=10, special="zero"),
             dict(n=300000, d=384, B=64, k=100, seed=5), dict(n=9000, d=384, B=5, k=1)]:
    try:
        allok &= case(**args)
    except Exception as e:
        allok = False
        print("EXC", args, repr(e))
print("ALL OK" if allok else "SOME FAILED")
