"""The CPU oracle against everything the reference pins on the search path (no GPU).

Reference known-answer test: lib/libmemex/src/storage/local.rs:175-214 (`test_hnsw`): three
3-d vectors, query [0.1,0.1,0.1], limit 3 -> 3 results, first id "test-two".  No reference test
asserts a score; the dist/score values below are the DistCosine / local.rs:86 arithmetic on those
vectors.
"""
import numpy as np
import pytest

from oracle import search_oracle as so

KAT_VECS = np.array([[0.0, 0.1, 0.2], [0.1, 0.1, 0.1], [0.3, 0.2, 0.1]], dtype=np.float32)
KAT_IDS = ["test-one", "test-two", "test-three"]
KAT_QUERY = np.array([0.1, 0.1, 0.1], dtype=np.float32)


def test_kat_rank_and_values(oracle):
    ids, dists, scores, nf = oracle.search(KAT_VECS, KAT_QUERY, 3)
    assert nf[0] == 3
    names = [KAT_IDS[int(i) - 1] for i in ids[0]]
    assert names[0] == "test-two"                        # the reference's own assertion
    assert names == ["test-two", "test-three", "test-one"]
    np.testing.assert_array_equal(dists[0], np.array([0.0, 0.0741799, 0.22540332], dtype=np.float32))
    np.testing.assert_array_equal(scores[0], np.array([1.0, 0.9258201, 0.7745967], dtype=np.float32))


def test_numpy_and_c_restatements_agree(oracle):
    rng = np.random.default_rng(7)
    for d in (1, 3, 17, 384):
        X = rng.standard_normal((50, d), dtype=np.float32)
        Q = rng.standard_normal((3, d), dtype=np.float32)
        i1, d1, s1 = so.search_np(X, Q, 5)
        i2, d2, s2, _ = oracle.search(X, Q, 5)
        np.testing.assert_array_equal(i1, i2)
        np.testing.assert_array_equal(d1.view(np.uint32), d2.view(np.uint32))
        np.testing.assert_array_equal(s1.view(np.uint32), s2.view(np.uint32))


def test_score_formula_edge_cases(oracle):
    # local.rs:86: dist 0 -> 1/0 = inf -> 1/inf = 0 -> score 1
    assert oracle.score(0.0) == np.float32(1.0)
    assert so.score_from_dist(np.float32(0.0)) == np.float32(1.0)
    for d in (1e-30, 1e-8, 0.25, 0.5, 1.0, 1.5, 2.0):
        assert oracle.score(d) == so.score_from_dist(np.float32(d))
        assert abs(float(oracle.score(d)) - (1.0 - d)) < 1e-6


def test_zero_norm_semantics(oracle):
    # DistCosine: either norm 0 -> distance 0 (so zero rows rank first, ties by id)
    X = np.array([[1, 0, 0], [0, 0, 0], [0.5, 0.5, 0], [0, 0, 0]], dtype=np.float32)
    ids, dists, scores, _ = oracle.search(X, np.array([1, 0, 0], dtype=np.float32), 4)
    assert ids[0].tolist() == [1, 2, 4, 3]
    assert dists[0, :3].tolist() == [0.0, 0.0, 0.0]
    ids, dists, _, _ = oracle.search(X, np.zeros(3, dtype=np.float32), 3)
    assert ids[0].tolist() == [1, 2, 3] and not dists[0].any()


def test_ties_break_by_id_and_clamp(oracle):
    base = np.array([0.3, -0.2, 0.9, 0.1], dtype=np.float32)
    X = np.stack([base * 2, base, base * 0.5, -base, base])
    ids, dists, _, _ = oracle.search(X, base, 5)
    assert ids[0, -1] == 4 and dists[0, -1] == np.float32(2.0)
    assert sorted(ids[0, :4].tolist()) == [1, 2, 3, 5]
    assert (dists[0] >= 0).all()


def test_fewer_rows_than_k_and_offsets(oracle):
    X = np.eye(3, dtype=np.float32)
    ids, dists, scores, nf = oracle.search(X, X[1], 5, id_offset=100)
    assert nf[0] == 3 and ids[0].tolist() == [102, 101, 103, 0, 0]
    assert np.isinf(dists[0, 3:]).all() and not scores[0, 3:].any()


def test_merge_equals_unsharded(oracle):
    rng = np.random.default_rng(3)
    X = rng.standard_normal((999, 32), dtype=np.float32)
    X[500:520] = X[3]                       # duplicates across the shard boundary
    Q = rng.standard_normal((7, 32), dtype=np.float32)
    Q[0] = X[3]
    full = oracle.search(X, Q, 10)
    parts = [(0, 333), (333, 700), (700, 999)]
    ids = np.stack([oracle.search(X[a:b], Q, 10, id_offset=a)[0] for a, b in parts])
    ds = np.stack([oracle.search(X[a:b], Q, 10, id_offset=a)[1] for a, b in parts])
    mi, md = oracle.merge(ids, ds)
    np.testing.assert_array_equal(mi, full[0])
    np.testing.assert_array_equal(md.view(np.uint32), full[1].view(np.uint32))


def test_golden_search_vectors(oracle):
    """tests/golden/search_golden.npz: committed outputs (generator: tests/golden/make_search_golden.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "search_golden.npz"))
    for name in [k[:-4] for k in g.files if k.endswith("_ids")]:
        n, d, B, k, seed = (int(x) for x in g[name + "_cfg"])
        rng = np.random.default_rng(seed)
        X = rng.standard_normal((n, d), dtype=np.float32)
        Q = rng.standard_normal((B, d), dtype=np.float32)
        ids, dists, scores, _ = oracle.search(X, Q, k)
        np.testing.assert_array_equal(ids, g[name + "_ids"])
        np.testing.assert_array_equal(dists.view(np.uint32), g[name + "_dists"].view(np.uint32))
        np.testing.assert_array_equal(scores.view(np.uint32), g[name + "_scores"].view(np.uint32))
