"""The in-library sharded index (mx_index_open_sharded, SURVEY.md section 8b `n_dev` / 8e) on a 1-GPU
box: logical shards on one device exercise routing (block-cyclic rows -> shards), the per-shard id
map, the exchange of the packed top-k blocks and the merge.  Bar: bit-identical to the oracle, i.e.
to the unsharded index, for every shard count."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import bits  # noqa: E402


def _same(idx, X, Q, k, oracle):
    oi, od, os_, onf = oracle.search(X, Q, k)
    ids, sc, di, nf = idx.search(Q, k)
    np.testing.assert_array_equal(ids, oi)
    np.testing.assert_array_equal(bits(di), bits(od))
    np.testing.assert_array_equal(bits(sc), bits(os_))
    np.testing.assert_array_equal(nf, onf)


@pytest.mark.parametrize("G,block_rows", [(2, 1024), (3, 64), (8, 4096), (8, 32)])
def test_logical_shards_equal_unsharded(G, block_rows, oracle, lib_built, tmp_path):
    from memex_amd.index import FlatIndex
    rng = np.random.default_rng(100 + G)
    X = rng.standard_normal((50003, 384), dtype=np.float32)
    X[777] = 0                                                   # zero-norm row
    X[30000:30020] = X[5]                                        # duplicates across shards
    Q = rng.standard_normal((19, 384), dtype=np.float32)
    Q[0] = X[5]
    Q[3] = 0
    with FlatIndex(384, devices=[0] * G, block_rows=block_rows) as idx:
        assert idx.n_shards == G
        assert idx.add(X[:1]) == 1
        assert idx.add(X[1:20001]) == 2                          # appends continue mid-block
        assert idx.add(X[20001:]) == 20002
        assert len(idx) == len(X)
        _same(idx, X, Q, 10, oracle)
        _same(idx, X, Q, 1, oracle)
        _same(idx, X, Q, 100, oracle)
        # the persisted file does not depend on the shard count: load it into a plain index
        idx.save(str(tmp_path))
        with FlatIndex(384) as plain:
            plain.load(str(tmp_path))
            assert len(plain) == len(X)
            _same(plain, X, Q, 10, oracle)
        # and back into a sharded one with another geometry
        with FlatIndex(384, devices=[0] * (G + 1), block_rows=96) as other:
            other.load(str(tmp_path))
            _same(other, X, Q, 10, oracle)
        idx.clear()
        assert len(idx) == 0
        assert idx.add(X[:100]) == 1                             # ids restart at 1 (local.rs:50,63)
        _same(idx, X[:100], Q, 10, oracle)


def test_sharded_device_api_and_small_shards(oracle, lib_built):
    """Device-pointer API on a composite; fewer rows than shards * k."""
    import torch
    from memex_amd.index import FlatIndex
    rng = np.random.default_rng(7)
    X = rng.standard_normal((37, 64), dtype=np.float32)
    Q = rng.standard_normal((5, 64), dtype=np.float32)
    with FlatIndex(64, devices=[0, 0, 0, 0], block_rows=32) as idx:
        idx.add_device(torch.from_numpy(X).cuda())
        k = 10
        ids = torch.zeros((5, k), dtype=torch.int64, device="cuda")
        sc = torch.zeros((5, k), device="cuda")
        di = torch.zeros((5, k), device="cuda")
        nf = torch.zeros((5,), dtype=torch.int32, device="cuda")
        idx.search_device(torch.from_numpy(Q).cuda(), k, ids, sc, di, nf)
        oi, od, os_, onf = oracle.search(X, Q, k)
        np.testing.assert_array_equal(ids.cpu().numpy().astype(np.uint64), oi)
        np.testing.assert_array_equal(bits(di.cpu().numpy()), bits(od))
        np.testing.assert_array_equal(bits(sc.cpu().numpy()), bits(os_))
        np.testing.assert_array_equal(nf.cpu().numpy(), onf)
        _same(idx, X, Q, 50, oracle)                             # k > n: n_found = n


def test_rccl_exchange_single_rank(oracle, lib_built):
    """MEMEX_HIP_EXCHANGE=rccl on one device: librccl is dlopen'ed, ncclCommInitAll(1 rank) and the
    all-gather run for real (the multi-device form of the same calls is what an 8-GPU host executes)."""
    from memex_amd.index import FlatIndex
    rng = np.random.default_rng(8)
    X = rng.standard_normal((20000, 384), dtype=np.float32)
    Q = rng.standard_normal((9, 384), dtype=np.float32)
    os.environ["MEMEX_HIP_EXCHANGE"] = "rccl"
    try:
        with FlatIndex(384, devices=[0]) as idx:
            idx.add(X)
            _same(idx, X, Q, 10, oracle)
    finally:
        del os.environ["MEMEX_HIP_EXCHANGE"]


def test_rccl_failure_at_search_time_falls_back_to_copies(oracle, lib_built):
    """A first multi-GPU run must not be lost to its collective: the communicator is self-tested when the index opens
    (one tiny all-gather under a deadline), and an all-gather that reports an error at search time
    (libmemex_hip_testing.so, the -DMEMEX_TESTING build, injects one) switches the index to peer copies for that batch and every
    later one -- same answers, `exchange` says so, the stats count it.  Run in a child process: the injection is once per process."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from memex_amd import _lib
_lib.use_testing_library()
from memex_amd.index import FlatIndex
from oracle.search_oracle import COracle
rng = np.random.default_rng(8)
X = rng.standard_normal((20000, 384), dtype=np.float32)
Q = rng.standard_normal((9, 384), dtype=np.float32)
oi, od, _, _ = COracle().search(X, Q, 10)
with FlatIndex(384, devices=[0]) as idx:
    assert idx.exchange == "rccl", idx.exchange
    idx.add(X)
    for _ in range(3):
        ids, sc, di, nf = idx.search(Q, 10)
        assert np.array_equal(ids, oi) and np.array_equal(di.view(np.uint32), od.view(np.uint32))
    assert idx.exchange == "p2p", idx.exchange
    assert idx.stats().exchange_fallbacks == 1
print("FALLBACK_OK")
'''
    env = dict(os.environ, MEMEX_HIP_EXCHANGE="rccl")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FALLBACK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "continues on peer copies" in r.stderr


def test_store_on_sharded_index(tmp_path, lib_built):
    """get_vector_storage(..., devices=[...]): the reference's store surface over the sharded index."""
    from memex_amd import storage
    from memex_amd.storage import VectorData
    rng = np.random.default_rng(9)
    vecs = rng.standard_normal((300, 16)).astype(np.float32)
    vs = storage.get_vector_storage(f"hip://{tmp_path}", "c", devices=[0, 0, 0])
    vs.add_vectors([VectorData(_id=f"seg-{i}", document_id="d", text="", vector=v, segment_id=i) for i, v in enumerate(vecs)])
    hits = vs.search(vecs[123], 3)
    assert hits[0][0] == "seg-123" and abs(hits[0][1] - 1.0) < 1e-6
    storage.evict_resident()
    vs2 = storage.get_vector_storage(f"hip://{tmp_path}", "c", devices=[0, 0, 0])   # reloaded from disk
    assert vs2.search(vecs[123], 3) == hits
    vs2.delete_collection()
    storage.evict_resident()


def test_helper_threads_on_logical_shards(oracle, lib_built):
    """MEMEX_HIP_SHARD_THREADS=1: the persistent per-shard helper threads (what shards on distinct devices
    always use) drive 8 logical shards of one GPU; many batches in a row, results as without them."""
    from memex_amd.index import FlatIndex
    rng = np.random.default_rng(10)
    X = rng.standard_normal((60000, 384), dtype=np.float32)
    Q = rng.standard_normal((300, 384), dtype=np.float32)        # two batches per call (256 + 44)
    os.environ["MEMEX_HIP_SHARD_THREADS"] = "1"
    try:
        with FlatIndex(384, devices=[0] * 8, block_rows=1024) as idx:
            assert idx.exchange == "p2p"
            idx.add(X)
            for _ in range(5):
                _same(idx, X, Q, 10, oracle)
    finally:
        del os.environ["MEMEX_HIP_SHARD_THREADS"]
    with FlatIndex(384) as plain:
        assert plain.exchange == "none"


def test_sharded_add_is_all_or_nothing(oracle, lib_built):
    """A device batch whose bad row lands on a LATER shard: the shards before it had already appended their
    part when the failure shows (device rows are validated by the ingest kernel) -- they must give it back,
    or later rows land at wrong offsets and searches return wrong ids."""
    import torch
    from memex_amd import _lib
    from memex_amd.index import FlatIndex
    rng = np.random.default_rng(11)
    X = rng.standard_normal((5000, 64), dtype=np.float32)
    X[100] = 0                                                    # a zero-norm row that stays
    bad = rng.standard_normal((4000, 64), dtype=np.float32)
    bad[50] = 0                                                   # a zero-norm row that must be rolled back (shard 0)
    bad[3000, 7] = np.nan                                         # rows 8000.. of the composite: block 7 -> shard 3
    Q = rng.standard_normal((7, 64), dtype=np.float32)
    with FlatIndex(64, devices=[0, 0, 0, 0], block_rows=1024) as idx:
        assert idx.add(X) == 1
        with pytest.raises(_lib.MemexHipError):
            idx.add_device(torch.from_numpy(bad).cuda())
        assert len(idx) == 5000
        _same(idx, X, Q, 10, oracle)
        more = rng.standard_normal((3000, 64), dtype=np.float32)
        assert idx.add_device(torch.from_numpy(more).cuda()) == 5001
        _same(idx, np.concatenate([X, more]), Q, 10, oracle)
        with pytest.raises(_lib.MemexHipError):                   # host rows: rejected before anything moves
            idx.add(bad)
        _same(idx, np.concatenate([X, more]), Q, 10, oracle)
