"""oracle/hnsw_baseline.cpp (bench.py's `cpu_baseline.hnsw`: the reference's real CPU algorithm with
memex's constants, local.rs:76,101) -- sanity on CPU: the reference's own known-answer test, exact
distances (DistCosine), and high recall where HNSW is known to work (low intrinsic dimension)."""
import numpy as np


def test_reference_kat_through_hnsw():
    from oracle.hnsw_baseline import HnswBaseline
    K = np.array([[0.0, 0.1, 0.2], [0.1, 0.1, 0.1], [0.3, 0.2, 0.1]], dtype=np.float32)   # local.rs:175-199
    h = HnswBaseline(K)
    ids, d, _ = h.search(np.array([[0.1, 0.1, 0.1]], dtype=np.float32), 3)
    assert ids[0].tolist() == [2, 3, 1]                                                   # "test-two" first (local.rs:211-212)
    np.testing.assert_array_equal(d[0], np.float32([0.0, 0.0741799, 0.22540332]))


def test_recall_and_distances_against_exact_search():
    from oracle.hnsw_baseline import EF_SEARCH, HnswBaseline
    from oracle.search_oracle import COracle
    rng = np.random.default_rng(0)
    n, d = 6000, 96
    Z = rng.standard_normal((n, 8), dtype=np.float32)
    P = rng.standard_normal((8, d), dtype=np.float32)
    X = (Z @ P + 0.05 * rng.standard_normal((n, d))).astype(np.float32)
    Q = (rng.standard_normal((32, 8), dtype=np.float32) @ P).astype(np.float32)
    h = HnswBaseline(X, seed=3, threads=4)
    ids, dist, sec = h.search(Q, 10, EF_SEARCH)
    oi, od, _, _ = COracle().search(X, Q, 10)
    rec = np.mean([len(set(a) & set(b)) / 10.0 for a, b in zip(ids.tolist(), oi.tolist())])
    assert rec >= 0.9 and sec > 0
    o = COracle()
    for b in range(4):                                                                    # reported distances are DistCosine's
        for j in range(10):
            assert dist[b, j] == o.dist(Q[b], X[int(ids[b, j]) - 1])
    assert (np.diff(dist, axis=1) >= 0).all()
