"""BASELINE.json full sizes on the GPU: cfg 3 (10M x 384 f32 in HBM, batch 256, top-10) and cfg 4's
per-GPU shard (10M x 768), wide rows (10M x 1024) on both filter copies, plus a clustered 1M corpus checked
against the oracle row for row.

The oracle cannot answer 10M x 256 in seconds, so the full-size checks are
 (a) size-independent properties (ordering, ranges, uniqueness, self-queries, idempotence),
 (b) AUTO == the library's all-f64 EXACT path bit for bit on a few queries (EXACT is oracle-checked at
     small sizes in test_search_gpu.py),
 (c) sharding invariance: the same corpus in 5 shards + the merge kernel == the unsharded answer,
 (d) the ORACLE on a row subset: for 8 queries, oracle distances of every reported row are bit-equal to
     the reported ones, and no row of a 1M-row strided subset (which contains every 10th row) beats
     or ties-with-lower-id the reported k-th neighbour without being reported.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import bits  # noqa: E402

B, K = 256, 10
BLOCK = 1_000_000


def _block(gen, b, d):
    import torch
    gen.manual_seed(1234 + b)
    return torch.randn((BLOCK, d), device="cuda", generator=gen)


def _build(n, d):
    import torch
    from memex_amd.index import FlatIndex
    idx = FlatIndex(d)
    idx.reserve(n)
    gen = torch.Generator(device="cuda")
    for b in range(n // BLOCK):
        xb = _block(gen, b, d)
        idx.add_device(xb)
        del xb
    gq = torch.Generator(device="cuda")
    gq.manual_seed(4321)
    q = torch.randn((B, d), device="cuda", generator=gq)
    # a few self-queries: row 12345 of block 3 and row 999999 of block 9
    q[0] = _block(gen, 3, d)[12345].clone()
    q[1] = _block(gen, 9, d)[999_999].clone() * 0.5
    torch.cuda.synchronize()
    return idx, q


def _search(idx, q, k):
    import torch
    ids = torch.zeros((q.shape[0], k), dtype=torch.int64, device="cuda")
    sc = torch.zeros((q.shape[0], k), dtype=torch.float32, device="cuda")
    di = torch.zeros((q.shape[0], k), dtype=torch.float32, device="cuda")
    nf = torch.zeros((q.shape[0],), dtype=torch.int32, device="cuda")
    idx.search_device(q, k, ids, sc, di, nf)
    return ids.cpu().numpy(), sc.cpu().numpy(), di.cpu().numpy(), nf.cpu().numpy()


def _properties(idx, q, n):
    from memex_amd import _lib
    assert len(idx) == n
    ids, sc, di, nf = _search(idx, q, K)
    st = idx.stats()
    assert (nf == K).all() and st.fallback_queries == 0 and st.retry_queries == 0
    assert (np.diff(di, axis=1) >= 0).all() and (di >= 0).all() and (di <= 2).all()
    assert all(len(set(r.tolist())) == K for r in ids)
    assert ids.min() >= 1 and ids.max() <= n
    np.testing.assert_allclose(sc, 1.0 - di, atol=1e-6)
    assert ids[0, 0] == 3_000_000 + 12345 + 1 and di[0, 0] <= 1e-7          # self queries
    assert ids[1, 0] == 9_000_000 + 999_999 + 1 and di[1, 0] <= 1e-7
    # AUTO == EXACT, bit for bit, on a handful of queries (EXACT = f64 arithmetic on every row)
    idx.set_search_mode(_lib.MX_SEARCH_EXACT)
    e_ids, e_sc, e_di, _ = _search(idx, q[:6].contiguous(), K)
    idx.set_search_mode(_lib.MX_SEARCH_AUTO)
    np.testing.assert_array_equal(ids[:6], e_ids)
    np.testing.assert_array_equal(bits(di[:6]), bits(e_di))
    np.testing.assert_array_equal(bits(sc[:6]), bits(e_sc))
    ids2, _, di2, _ = _search(idx, q, K)                                      # idempotence
    np.testing.assert_array_equal(ids, ids2)
    np.testing.assert_array_equal(di, di2)
    return ids, sc, di


def _oracle_on_subset(oracle, ids, di, q, n, d, nq=8, every=10):
    """(d) of the module docstring."""
    import torch
    gen = torch.Generator(device="cuda")
    qh = q[:nq].cpu().numpy()
    want_rows = np.unique(ids[:nq].astype(np.int64) - 1)
    reported, subset = {}, []
    for b in range(n // BLOCK):
        xb = _block(gen, b, d)
        sel = want_rows[(want_rows >= b * BLOCK) & (want_rows < (b + 1) * BLOCK)]
        for r in sel:
            reported[int(r)] = xb[int(r) - b * BLOCK].cpu().numpy()
        subset.append(xb[::every].cpu().numpy())
        del xb
    subset = np.concatenate(subset)                                           # global rows 0, every, 2*every, ...
    for b in range(nq):
        rows = np.stack([reported[int(r) - 1] for r in ids[b]])
        np.testing.assert_array_equal(bits(oracle.all_dists(rows, qh[b])), bits(di[b]))   # reported dists are the oracle's
        sd = oracle.all_dists(subset, qh[b])
        sid = np.arange(subset.shape[0], dtype=np.int64) * every + 1
        kth_d, kth_id = di[b, -1], int(ids[b, -1])
        better = (sd < kth_d) | ((sd == kth_d) & (sid < kth_id))
        missing = set(sid[better].tolist()) - set(int(x) for x in ids[b])
        assert not missing, f"query {b}: subset rows {sorted(missing)[:5]} beat the reported k-th neighbour"


def _sharding_invariance(ids, di, q, n, d, shards=5):
    """(c): the same rows in `shards` indexes with global id offsets + the merge kernel."""
    import torch
    from memex_amd.index import FlatIndex, merge_topk_packed_device, packed_result_block
    gen = torch.Generator(device="cuda")
    per = n // shards
    assert per % BLOCK == 0
    g_block = torch.zeros((shards, B * K * 12), dtype=torch.uint8, device="cuda")
    for s in range(shards):
        with FlatIndex(d) as idx:
            idx.reserve(per)
            idx.set_id_offset(s * per)
            for b in range(s * per // BLOCK, (s + 1) * per // BLOCK):
                xb = _block(gen, b, d)
                idx.add_device(xb)
                del xb
            block, s_ids, s_di = packed_result_block(B, K, "cuda")
            sc = torch.zeros((B, K), device="cuda")
            nf = torch.zeros((B,), dtype=torch.int32, device="cuda")
            idx.search_device(q, K, s_ids, sc, s_di, nf)
            g_block[s] = block
        torch.cuda.empty_cache()
    m_ids = torch.zeros((B, K), dtype=torch.int64, device="cuda")
    m_di = torch.zeros((B, K), device="cuda")
    m_sc = torch.zeros((B, K), device="cuda")
    torch.cuda.synchronize()
    merge_topk_packed_device(0, g_block, shards, B, K, m_ids, m_di, m_sc)
    np.testing.assert_array_equal(m_ids.cpu().numpy(), ids)
    np.testing.assert_array_equal(bits(m_di.cpu().numpy()), bits(di))


def test_cfg3_10m_x_384(oracle, lib_built):
    import torch
    n, d = 10_000_000, 384
    idx, q = _build(n, d)
    try:
        ids, sc, di = _properties(idx, q, n)
        _oracle_on_subset(oracle, ids, di, q, n, d)
    finally:
        idx.close()
        torch.cuda.empty_cache()
    _sharding_invariance(ids, di, q, n, d)


def test_cfg4_shard_10m_x_768(oracle, lib_built):
    """BASELINE configs[3]: 80M x 768 over 8 GPUs = 10M x 768 per GPU (30.7 GB f32 + 15.4 GB filter
    copy): the dim_pad = 768 kernels (KC = 6) at the per-GPU size they run at."""
    import torch
    n, d = 10_000_000, 768
    idx, q = _build(n, d)
    try:
        ids, sc, di = _properties(idx, q, n)
        _oracle_on_subset(oracle, ids, di, q, n, d, nq=4, every=20)
    finally:
        idx.close()
        torch.cuda.empty_cache()


def test_wide_rows_10m_x_1024(oracle, lib_built):
    """Wide rows at full size (bge-large / e5-large width; 41 GB of f32 rows): the int8 scan (the library's choice at
    1024 dims: 256 queries per pass, KC = 8) and, on the same index, the bf16 copy through scan16w_kernel (the k-steps
    of a row dealt to two waves, 128 queries per pass) -- same properties, same oracle check, same answers."""
    import torch
    n, d = 10_000_000, 1024
    idx, q = _build(n, d)
    try:
        assert idx.stats().filter_kind == 2
        ids, sc, di = _properties(idx, q, n)
        _oracle_on_subset(oracle, ids, di, q, n, d, nq=4, every=40)
        idx.set_filter_copy("bf16")
        idx.reset_stats()
        ids2, sc2, di2 = _properties(idx, q, n)
        st = idx.stats()
        assert st.filter_kind == 3 and st.scan_launches >= 4          # two passes per 256-query batch
        np.testing.assert_array_equal(ids2, ids)
        np.testing.assert_array_equal(bits(di2), bits(di))
        np.testing.assert_array_equal(bits(sc2), bits(sc))
    finally:
        idx.close()
        torch.cuda.empty_cache()


def test_clustered_1m_against_the_oracle(oracle, lib_built):
    """Dense neighbourhoods (20k clusters, intra-cluster cosine 0.8 .. 0.95, 1 % exact duplicates, rows of
    random length): the bf16 filter keeps thousands of candidates per query here, not dozens.
    Row-for-row oracle parity on 1M rows, and no query may leave the fast path."""
    import torch
    import bench
    from memex_amd.index import FlatIndex
    n, d, nq = 1_000_000, 384, 24
    cen = bench.clustered_centres(d)
    x = bench.clustered_rows(n, d, 5000, cen)
    q = bench.clustered_rows(nq, d, 4321, cen)
    q[3] = x[777_777] * 1.5                                                    # a query that IS a (duplicated?) row
    with FlatIndex(d) as idx:
        idx.add_device(x)
        ids, sc, di, nf = _search(idx, q, K)
        st = idx.stats()
    xh, qh = x.cpu().numpy(), q.cpu().numpy()
    del x
    torch.cuda.empty_cache()
    oi, od, os_, onf = oracle.search(xh, qh, K)
    np.testing.assert_array_equal(ids.astype(np.uint64), oi)
    np.testing.assert_array_equal(bits(di), bits(od))
    np.testing.assert_array_equal(bits(sc), bits(os_))
    np.testing.assert_array_equal(nf, onf)
    assert st.fallback_queries == 0
