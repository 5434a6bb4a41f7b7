"""The CENTRED bf16 filter copy (scan16.hip / launch_shadow): a bf16 copy rebuilt from a populated
index whose rows sit in a cone holds r_c = c/|c| - a_c m (m = the rows' mean direction) plus a_c per row; the scan scores
a_q a_c + r_q . r_c.  The residual vectors are several times shorter than the unit rows, and so is the rounding error the
certificate covers -- corpora that used to overflow into retry passes and the EXACT path (embeddings of one model: a narrow
cone) are answered by the first pass.  Results stay bit-identical to the oracle (reference arithmetic: local.rs:71-91 +
DistCosine); what is checked on top is that the certificate holds and that it did get tighter."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def cone_rows(rng, n, d, spread=0.2, axis_seed=1):
    """rows = axis + spread * noise (unit-variance total): mean pairwise cosine 1 / (1 + spread^2) ~ 0.96 at 0.2"""
    axis = np.random.default_rng(axis_seed).standard_normal(d).astype(np.float32)
    axis /= np.linalg.norm(axis)
    x = axis[None, :] + (spread / np.sqrt(d)) * rng.standard_normal((n, d), dtype=np.float32)
    return (x * rng.uniform(0.5, 2.0, (n, 1)).astype(np.float32)).astype(np.float32)   # random lengths: the path normalises


def _equal(idx, X, Q, k, oracle):
    oi, od, os_, onf = oracle.search(X, Q, k)
    ids, sc, di, nf = idx.search(Q, k)
    np.testing.assert_array_equal(ids, oi)
    np.testing.assert_array_equal(bits(di), bits(od))
    np.testing.assert_array_equal(bits(sc), bits(os_))
    np.testing.assert_array_equal(nf, onf)


@pytest.mark.parametrize("n,d,B,seed", [(50000, 384, 64, 1), (30000, 768, 256, 2), (20000, 100, 33, 3), (40000, 512, 40, 4)])
def test_centred_copy_certificate_and_parity(n, d, B, seed, oracle, lib_built):
    from memex_amd.index import FlatIndex
    rng = np.random.default_rng(seed)
    X = cone_rows(rng, n, d)
    X[17] = 0.0                                    # a zero-norm row (stored as zeros, a_c = 0, listed)
    X[100:140] = X[99]                             # exact duplicates
    Q = cone_rows(rng, B, d)
    Q[0] = X[99]
    with FlatIndex(d) as idx:
        idx.add(X[: n - 5000])
        idx.set_filter_copy("bf16")                # rebuilt from the resident rows: centred
        assert idx.stats().filter_centred == 1 and idx.stats().filter_kind == 3
        idx.set_profiling(True)
        _equal(idx, X[: n - 5000], Q, 10, oracle)
        st = idx.stats()
        assert st.fallback_queries == 0 and st.retry_queries == 0
        # the certificate holds, and it is several times tighter than a plain bf16 copy's (0.0041-0.0045 on unit rows)
        assert 0.0 < st.max_abs_err <= st.approx_err_bound <= 0.0016, (st.max_abs_err, st.approx_err_bound)
        # appends after the centre was fixed: a partly filled tile, then growth past the capacity (the a_c array moves along)
        idx.add(X[n - 5000: n - 4990])
        idx.add(X[n - 4990:])
        assert idx.stats().filter_centred == 1
        _equal(idx, X, Q, 10, oracle)
        _equal(idx, X, Q[:3], 100, oracle)
        big = cone_rows(rng, 3 * n, d)
        idx.add(big)
        assert idx.stats().filter_centred == 1
        XX = np.concatenate([X, big])
        _equal(idx, XX, Q[:16], 10, oracle)
        assert idx.stats().fallback_queries == 0
        idx.clear()
        assert idx.stats().filter_centred == 0


@pytest.mark.parametrize("n,d,B,seed", [(50000, 384, 64, 21), (40000, 768, 256, 22), (20000, 100, 33, 23), (40000, 512, 300, 24),
                                        (40000, 256, 50, 25), (40000, 640, 128, 26)])   # (768: > 2 x 256 scan tiles, so that the sample pass runs; 256 / 640 dims: the two-slot and five-slot centred kernels)
def test_centred_int8_copy_certificate_and_parity(n, d, B, seed, oracle, lib_built):
    """Round 6: the same split for the int8 copy.  shadow8_kernel quantises r_c = c/|c| - a_c m (its step and its
    residual bound shrink with the vector), a_c rides in scan8_kernel's DMA stream (256 bytes per 64-row tile), the tile epilogue
    scores a_q a_c + s_h s_q sum per row, and a row's bound is Eq |r_c|max + (|r_q| + Eq) e_h.  Ids / dists / scores stay the
    oracle's bits; the certificate holds and is an order of magnitude tighter than the plain int8 copy's 0.022-0.027."""
    from memex_amd.index import FlatIndex
    rng = np.random.default_rng(seed)
    X = cone_rows(rng, n, d)
    X[17] = 0.0
    X[100:140] = X[99]
    Q = cone_rows(rng, B, d)
    Q[0] = X[99]
    Q[1] = rng.standard_normal(d).astype(np.float32)      # a query off the cone (a_q ~ 0, |r_q| ~ 1)
    with FlatIndex(d) as idx:
        idx.add(X[: n - 5000])
        idx.set_filter_copy("bf16")
        idx.set_filter_copy("i8")                           # rebuilt from the resident rows: centred
        st = idx.stats()
        assert st.filter_centred == 1 and st.filter_kind == 2
        idx.set_profiling(True)
        _equal(idx, X[: n - 5000], Q, 10, oracle)
        st = idx.stats()
        assert st.fallback_queries == 0 and st.retry_queries == 0 and st.filter_kind == 2
        assert 0.0 < st.max_abs_err <= st.approx_err_bound <= 0.008, (st.max_abs_err, st.approx_err_bound)   # (the off-cone query's bound)
        # appends after the centre was fixed: a partly filled half tile is requantised as a whole, then growth past the capacity
        idx.add(X[n - 5000: n - 4990])
        idx.add(X[n - 4990:])
        assert idx.stats().filter_centred == 1
        _equal(idx, X, Q, 10, oracle)
        _equal(idx, X, Q[:3], 100, oracle)
        big = cone_rows(rng, 3 * n, d)
        idx.add(big)
        st = idx.stats()
        assert st.filter_centred == 1 and st.filter_kind == 2
        XX = np.concatenate([X, big])
        _equal(idx, XX, Q[:16], 10, oracle)
        assert idx.stats().fallback_queries == 0
        idx.clear()
        assert idx.stats().filter_centred == 0


def test_certificate_of_the_centred_int8_copy_on_cone_queries(oracle, lib_built):
    """Queries from inside the cone (what an encoder produces): the bound of the worst half tile stays below 0.004 on rows whose
    residual vectors are 0.2 long (plain int8 copy: 0.022-0.027), and the measured error below the bound."""
    from memex_amd.index import FlatIndex
    rng = np.random.default_rng(31)
    d, n = 384, 60000
    X = cone_rows(rng, n, d)
    Q = cone_rows(rng, 256, d)
    with FlatIndex(d) as idx:
        idx.add(X)
        idx.set_filter_copy("bf16")
        idx.set_filter_copy("i8")
        assert idx.stats().filter_centred == 1 and idx.stats().filter_kind == 2
        idx.set_profiling(True)
        _equal(idx, X, Q, 10, oracle)
        st = idx.stats()
        assert 0.0 < st.max_abs_err <= st.approx_err_bound <= 0.004, (st.max_abs_err, st.approx_err_bound)
        assert st.fallback_queries == 0 and st.retry_queries == 0


def test_rows_without_a_cone_stay_uncentred(oracle, lib_built):
    from memex_amd.index import FlatIndex
    rng = np.random.default_rng(5)
    X = rng.standard_normal((20000, 384), dtype=np.float32)
    Q = rng.standard_normal((8, 384), dtype=np.float32)
    with FlatIndex(384) as idx:
        idx.add(X)
        idx.set_filter_copy("bf16")
        assert idx.stats().filter_centred == 0     # |mean of the unit rows| ~ 0.007: nothing to remove
        _equal(idx, X, Q, 10, oracle)
        idx.clear()
        idx.add(cone_rows(rng, 20000, 384))
        idx.set_filter_copy("i8")
        idx.set_filter_copy("bf16")
        assert idx.stats().filter_centred == 1
        idx.clear()                                # the centre goes with the rows it was computed from
        assert idx.stats().filter_centred == 0
        idx.add(X)
        _equal(idx, X, Q, 10, oracle)


def test_narrow_cone_1m_is_answered_without_the_exact_path(oracle, lib_built):
    """1M rows in a narrow cone (mean pairwise cosine >= 0.95; every row has ~100 near copies at cosine 0.99:
    the shape of bench.py's enc_like leg), the library's own choice of copy: the PLAIN int8 certificate cannot resolve it, the copy
    is rebuilt ONCE, centred on the rows' mean direction -- still int8, no demotion (round 5 went to a centred bf16 copy, twice
    the bytes) -- and from then on no query needs the retry pass or the EXACT path; ids / dists / scores equal the oracle's bit
    for bit."""
    from memex_amd.index import FlatIndex
    rng = np.random.default_rng(6)
    d, n_src, n = 384, 10000, 1_000_000
    src = cone_rows(rng, n_src, d, spread=0.2)
    src /= np.linalg.norm(src, axis=1, keepdims=True)
    pick = rng.integers(0, n_src, n)
    X = (src[pick] + (0.1 / np.sqrt(d)) * rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    Q = cone_rows(rng, 256, d, spread=0.2)
    sub = X[:2000] / np.linalg.norm(X[:2000], axis=1, keepdims=True)
    assert (sub @ sub.T).mean() >= 0.95
    with FlatIndex(d) as idx:
        idx.add(X)
        first = idx.search(Q, 10)
        st0 = idx.stats()
        assert st0.filter_demotions == 0 and st0.filter_kind == 2 and st0.filter_centred == 1, (st0.filter_demotions, st0.filter_kind, st0.filter_centred)
        idx.reset_stats()
        ids, sc, di, nf = idx.search(Q, 10)
        st = idx.stats()
        assert st.fallback_queries == 0 and st.retry_queries == 0, (st.fallback_queries, st.retry_queries)
        assert st.candidates / 256 < 2000
        oi, od, os_, onf = oracle.search(X, Q[:24], 10)
        for got in (first, (ids, sc, di, nf)):
            np.testing.assert_array_equal(got[0][:24], oi)
            np.testing.assert_array_equal(bits(got[2][:24]), bits(od))
            np.testing.assert_array_equal(bits(got[1][:24]), bits(os_))


def test_sharded_index_over_a_cone_corpus(oracle, lib_built):
    """Every shard of the in-library sharded index judges (and demotes, and centres) its own copy: three logical shards over a
    narrow-cone corpus still answer with the oracle's bits, and none of them needs the EXACT path after its demotion."""
    from memex_amd.index import FlatIndex
    rng = np.random.default_rng(8)
    d, n_src, n = 384, 3000, 240_000
    src = cone_rows(rng, n_src, d)
    src /= np.linalg.norm(src, axis=1, keepdims=True)
    X = (src[rng.integers(0, n_src, n)] + (0.1 / np.sqrt(d)) * rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    Q = cone_rows(rng, 64, d)
    oi, od, os_, onf = oracle.search(X, Q, 10)
    with FlatIndex(d, devices=[0, 0, 0], block_rows=4096) as idx:
        idx.add(X)
        for _ in range(2):
            ids, sc, di, nf = idx.search(Q, 10)
            np.testing.assert_array_equal(ids, oi)
            np.testing.assert_array_equal(bits(di), bits(od))
            np.testing.assert_array_equal(bits(sc), bits(os_))
        idx.reset_stats()
        idx.search(Q, 10)
        st = idx.stats()
        assert st.fallback_queries == 0 and st.filter_centred == 1, (st.fallback_queries, st.filter_centred, st.filter_kind)


def test_centred_copy_with_out_of_range_norms_and_queries_off_the_cone(oracle, lib_built):
    """What rides on the side lists of the plain copies must ride on them under a centred copy too: rows whose norm is outside
    the f32 stages' range (stored as zeros, evaluated in f64 for every query), zero rows, and queries that have nothing to do
    with the cone (a_q ~ 0), are scaled by 1e-30 / 1e30, or are a row of the corpus."""
    from memex_amd.index import FlatIndex
    rng = np.random.default_rng(12)
    d, n = 384, 30000
    X = cone_rows(rng, n, d)
    X[5] *= np.float32(1e-19)                      # norms far outside [1e-15, 1e15]: the wild list
    X[6] *= np.float32(1e18)
    X[7] *= np.float32(3e-16)
    X[50] = 0.0
    X[51] = 0.0
    Q = cone_rows(rng, 40, d)
    Q[1] = rng.standard_normal(d).astype(np.float32)                # off the cone
    Q[2] = -Q[3]                                                      # opposite the cone: every cosine negative
    Q[4] = Q[4] * np.float32(1e-30)
    Q[5] = Q[5] * np.float32(1e15)                                   # (f32 self-products still finite: 1e30 would make DistCosine's norm inf)
    Q[6] = X[6]                                                       # the huge row itself
    Q[7] = X[5]                                                       # the tiny row itself
    with FlatIndex(d) as idx:
        idx.add(X)
        idx.set_filter_copy("bf16")
        assert idx.stats().filter_centred == 1
        for k in (1, 10, 64):
            _equal(idx, X, Q, k, oracle)
        # a query whose f32 self-products overflow (|q_i| > 1.8e19): DistCosine's norm is inf, every non-zero row is at
        # distance 1 - dot/inf = 1, and a row whose dot product overflows too makes 1 - inf/inf = NaN, where the reference
        # asserts (panics).  Outside the arithmetic's domain the call still returns: zero-norm rows (distance 0) first.
        Qinf = (Q[8:9] * np.float32(1e30)).astype(np.float32)
        ids, sc, di, nf = idx.search(Qinf, 5)
        assert nf[0] == 5 and sorted(ids[0, :2].tolist()) == [51, 52] and di[0, 0] == 0.0 and di[0, 2] == 1.0
        # rows appended behind the centred copy, all of them with norms outside the range (up to the list's capacity)
        extra = (cone_rows(np.random.default_rng(13), 900, d) * np.float32(1e-17)).astype(np.float32)
        idx.add(extra)
        _equal(idx, np.concatenate([X, extra]), Q, 10, oracle)


def test_a_pinned_bf16_copy_that_started_small_is_centred_once_the_collection_has_grown(oracle, lib_built):
    """A bf16 copy that grew with its index from empty (or was built when the collection held < 256 rows) is plain.  Once
    the collection has doubled it is rebuilt -- and comes back centred if the rows sit in a cone by then; rows without a cone
    are looked at again at every doubling and stay plain.  Answers bit-identical throughout."""
    from memex_amd.index import FlatIndex
    rng = np.random.default_rng(21)
    d = 384
    X = cone_rows(rng, 1500, d)
    Q = cone_rows(rng, 20, d)
    with FlatIndex(d) as idx:
        idx.set_filter_copy("bf16")                    # pinned, on an empty index
        idx.add(X[:100])
        idx.add(X[100:200])
        assert idx.stats().filter_centred == 0         # 200 rows: too few to judge
        _equal(idx, X[:200], Q, 5, oracle)
        idx.add(X[200:300])                            # 300 rows >= 256: rebuilt, and these rows sit in a cone
        assert idx.stats().filter_centred == 1
        _equal(idx, X[:300], Q, 10, oracle)
        idx.add(X[300:])                               # appended behind the centred copy
        assert idx.stats().filter_centred == 1
        _equal(idx, X, Q, 10, oracle)
        idx.clear()
        G = rng.standard_normal((900, d), dtype=np.float32)
        idx.add(G[:300])
        assert idx.stats().filter_centred == 0         # looked at (>= 256 rows since the clear), no cone: plain
        idx.add(G[300:])                               # doubled: looked at again, still plain
        assert idx.stats().filter_centred == 0 and idx.stats().filter_kind == 3
        _equal(idx, G, rng.standard_normal((9, d), dtype=np.float32), 10, oracle)


@pytest.mark.parametrize("seed", list(range(40, 52)))
def test_centred_int8_copy_random_cones(seed, oracle, lib_built):
    """Random cone geometry against the oracle, bit for bit: row width (every centred kernel, 1 .. 6 slots of 128 dims), cone tightness
    from 0.02 (residual steps near the floor kMinStep8) to 1.0 (hardly a cone), batch size, queries inside the cone, off it, and on
    its FAR side (a_q < 0: the accumulators' initial values are negative and truncate toward zero), rows that are exactly the axis
    (zero residual), duplicates, a zero row, k = 10 and k = 64.  Whatever the certificate does with such a corpus -- first pass, retry,
    demotion -- the answers are the oracle's."""
    from memex_amd.index import FlatIndex
    rng = np.random.default_rng(seed)
    d = int(rng.choice([64, 128, 200, 256, 384, 448, 512, 600, 640, 768]))
    n = int(rng.integers(12000, 30000))
    B = int(rng.integers(1, 200))
    spread = float(np.exp(rng.uniform(np.log(0.02), np.log(1.0))))
    X = cone_rows(rng, n, d, spread=spread, axis_seed=seed)
    axis = np.random.default_rng(seed).standard_normal(d).astype(np.float32)
    axis /= np.linalg.norm(axis)
    X[5] = 0.0
    X[6:12] = axis * 3.0                                   # rows ON the axis: zero residual
    X[200:230] = X[199]
    Q = cone_rows(rng, B, d, spread=spread, axis_seed=seed)
    if B > 1:
        Q[1] = -Q[1]                                       # the far side of the cone: a_q ~ -1
    if B > 2:
        Q[2] = rng.standard_normal(d).astype(np.float32)   # off the cone
    if B > 3:
        Q[3] = axis                                        # the axis itself
    with FlatIndex(d) as idx:
        idx.add(X)
        idx.set_filter_copy("bf16")
        idx.set_filter_copy("i8")                          # rebuilt from the resident rows: centred when they sit in a cone
        _equal(idx, X, Q, 10, oracle)
        _equal(idx, X, Q[: min(B, 8)], 64, oracle)
        more = cone_rows(rng, 3000, d, spread=spread, axis_seed=seed)
        idx.add(more)
        _equal(idx, np.concatenate([X, more]), Q[: min(B, 32)], 10, oracle)
