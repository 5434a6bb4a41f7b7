"""N>1 wiring on CPU: world_size-2 gloo processes drive ShardedFlatIndex (partition, global id
offsets, all-gather, merge).  The local index and the merge are injected test doubles backed by
the oracle -- the product defaults are the HIP index and the HIP merge kernel (GPU tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class OracleLocalIndex:
    """Test double with FlatIndex's surface, computing on the CPU oracle."""

    def __init__(self, dim):
        from oracle.search_oracle import COracle
        self.orc = COracle()
        self.dim = dim
        self.rows = np.zeros((0, dim), dtype=np.float32)
        self.off = 0

    def reserve(self, n):
        pass

    def set_id_offset(self, off):
        self.off = off

    def add(self, rows):
        self.rows = np.concatenate([self.rows, np.asarray(rows, dtype=np.float32)])
        return self.off + 1

    def search_device(self, q, k, ids, scores, dists, nf):
        i, d, s, n = self.orc.search(self.rows, q.numpy(), k, id_offset=self.off)
        ids.copy_(torch.from_numpy(i.astype(np.int64)))
        dists.copy_(torch.from_numpy(d))
        scores.copy_(torch.from_numpy(s))
        nf.copy_(torch.from_numpy(n))

    def close(self):
        pass


def _oracle_merge(g_ids, g_dists, m_ids, m_dists, m_scores):
    from oracle.search_oracle import COracle, score_from_dist
    oi, od = COracle().merge(g_ids.numpy().astype(np.uint64), g_dists.numpy())
    m_ids.copy_(torch.from_numpy(oi.astype(np.int64)))
    m_dists.copy_(torch.from_numpy(od))
    m_scores.copy_(torch.from_numpy(score_from_dist(od) * (oi != 0)))


def _worker(rank, world, port, n, d, B, k, q_out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from memex_amd.sharded import ShardedFlatIndex
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, d), dtype=np.float32)
    X[n // 2 - 3: n // 2 + 3] = X[1]                    # duplicates straddling the shard boundary
    Q = rng.standard_normal((B, d), dtype=np.float32)
    Q[0] = X[1]
    sh = ShardedFlatIndex(d, n, index_factory=lambda: OracleLocalIndex(d), merge_fn=_oracle_merge)
    assert (sh.rank, sh.world) == (rank, world)
    sh.add_local(X[sh.lo:sh.hi])
    ids, dists, scores = sh.search(torch.from_numpy(Q), k)
    if rank == 0:
        q_out.put((ids.numpy(), dists.numpy(), scores.numpy()))
    # every rank holds the same answer
    ref = [torch.zeros_like(ids) for _ in range(world)]
    dist.all_gather(ref, ids)
    assert all(torch.equal(r, ref[0]) for r in ref)
    dist.destroy_process_group()


def test_two_rank_sharded_search_equals_unsharded():
    from oracle.search_oracle import COracle
    n, d, B, k = 501, 24, 5, 10
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, d, B, k, q)) for r in range(2)]
    for p in procs:
        p.start()
    ids, dists, scores = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, d), dtype=np.float32)
    X[n // 2 - 3: n // 2 + 3] = X[1]
    Q = rng.standard_normal((B, d), dtype=np.float32)
    Q[0] = X[1]
    fi, fd, fs, _ = COracle().search(X, Q, k)
    np.testing.assert_array_equal(ids.astype(np.uint64), fi)
    np.testing.assert_array_equal(dists.view(np.uint32), fd.view(np.uint32))
    np.testing.assert_array_equal(scores.view(np.uint32), fs.view(np.uint32))
