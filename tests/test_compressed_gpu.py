"""Compressed corpus (MX_CORPUS_BF16): the index keeps only bf16(c/|c|).
Bar: searches are EXACT with respect to the stored rows -- ids, dists and scores bit-identical to the
oracle (the reference's arithmetic) applied to mx_index_get_rows() -- and the stored rows are the
inserted ones to bf16 precision."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import bits  # noqa: E402


def _check_against_stored_rows(idx, Q, k, oracle):
    rows = idx.get_rows(0, len(idx))
    oi, od, os_, onf = oracle.search(rows, Q, k)
    ids, sc, di, nf = idx.search(Q, k)
    np.testing.assert_array_equal(ids, oi)
    np.testing.assert_array_equal(bits(di), bits(od))
    np.testing.assert_array_equal(bits(sc), bits(os_))
    np.testing.assert_array_equal(nf, onf)
    return rows, ids


@pytest.mark.parametrize("n,d,seed", [(50003, 384, 1), (20000, 768, 2), (7001, 100, 3), (40, 384, 4), (20000, 1024, 5), (9000, 1500, 6)])
def test_compressed_search_is_exact_on_the_stored_rows(n, d, seed, oracle, lib_built, tmp_path):
    from memex_amd import _lib
    from memex_amd.index import FlatIndex
    rng = np.random.default_rng(seed)
    X = (rng.standard_normal((n, d)) * rng.uniform(0.2, 5.0, (n, 1))).astype(np.float32)
    Q = rng.standard_normal((21, d), dtype=np.float32)
    if n > 1000:
        X[5] = 0                                                      # zero-norm row: dist 0 to everything
        X[900:920] = X[17]                                            # exact duplicates
        Q[0] = X[17] * 3
        Q[2] = 0
    with FlatIndex(d) as idx:
        idx.set_corpus_mode("bf16")
        cuts = [0, 1, min(n, 45), min(n, 4133), n]                    # appends that start and end inside tiles
        for a, b in zip(cuts, cuts[1:]):
            if b > a:
                assert idx.add(X[a:b]) == a + 1
        assert len(idx) == n
        st = idx.stats()
        assert st.filter_copy_bytes > 0
        rows, ids = _check_against_stored_rows(idx, Q, 10, oracle)
        _check_against_stored_rows(idx, Q, 1, oracle)
        _check_against_stored_rows(idx, Q, 64, oracle)
        # the stored rows are the inserted ones, normalised, to bf16 precision
        nrm = np.linalg.norm(X, axis=1, keepdims=True)
        unit = np.divide(X, nrm, out=np.zeros_like(X), where=nrm > 0)
        assert np.abs(rows - unit).max() <= 2.0 ** -8 * np.abs(unit).max() + 1e-6
        if n > 1000:
            assert not rows[5].any()
            # recall against the f32 corpus (sanity: each cosine moved by < 2e-3)
            fi = oracle.search(X, Q, 10)[0]
            rec = np.mean([len(set(a) & set(b)) / 10.0 for a, b in zip(ids.tolist(), fi.tolist())])
            assert rec >= 0.9
        # EXACT mode reads the stored rows too
        idx.set_search_mode(_lib.MX_SEARCH_EXACT)
        _check_against_stored_rows(idx, Q[:3], 10, oracle)
        idx.set_search_mode(_lib.MX_SEARCH_AUTO)
        # persistence: the stored values come back unchanged, into either corpus mode
        idx.save(str(tmp_path))
        with FlatIndex(d) as again:
            again.set_corpus_mode("bf16")
            again.load(str(tmp_path))
            np.testing.assert_array_equal(bits(again.get_rows(0, n)), bits(rows))
            np.testing.assert_array_equal(again.search(Q, 10)[0], ids)
        with FlatIndex(d) as plain:
            plain.load(str(tmp_path))                                 # f32 index holding the same (rounded) rows
            np.testing.assert_array_equal(bits(plain.get_rows(0, n)), bits(rows))
            np.testing.assert_array_equal(plain.search(Q, 10)[0], ids)
        with pytest.raises(_lib.MemexHipError):
            idx.set_corpus_mode("f32")                                # only while empty
        with pytest.raises(_lib.MemexHipError):
            idx.set_filter_copy(False)                                # there is no f32 copy to fall back to


def test_compressed_sharded_and_clustered(oracle, lib_built):
    import torch
    import bench
    from memex_amd.index import FlatIndex
    n, d = 200_000, 384
    cen = bench.clustered_centres(d)
    x = bench.clustered_rows(n, d, 5000, cen)
    q = bench.clustered_rows(16, d, 4321, cen).cpu().numpy()
    with FlatIndex(d, devices=[0, 0, 0], block_rows=4096) as idx:
        idx.set_corpus_mode("bf16")
        idx.add_device(x)
        del x
        torch.cuda.empty_cache()
        _check_against_stored_rows(idx, q, 10, oracle)
        assert idx.stats().fallback_queries == 0
