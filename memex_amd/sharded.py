"""Row-sharded multi-GPU flat index: one process per GPU, RCCL all-gather of local top-k.

SURVEY.md section 8(e): the corpus is split by contiguous row ranges, GPU g owns rows
``[g*N/G, (g+1)*N/G)`` and reports GLOBAL ids (``id_offset + local_row + 1``, the reference's dense
1-based insertion ids, lib/libmemex/src/storage/local.rs:63).  Every rank answers the same query
batch on its shard; ONE collective -- an all-gather of the per-shard ``(ids, dists)`` lists
(B*k*12 bytes per rank; latency-bound over xGMI) -- followed by a merge kernel ordered by
``(dist, id)`` gives every rank the global answer, bit-identical for any G.

``torch.distributed`` backend ``"nccl"`` is RCCL on ROCm.  The reference has no distributed path at
all (single process); this module is new surface, shaped like ``VectorStore.search``.

The local index and the merge function are injectable so that the wiring (partition, offsets,
collective, merge) can be exercised on CPU with the ``gloo`` backend in tests; the defaults are the
HIP index and the HIP merge kernel -- there is no CPU implementation in the product.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import torch
import torch.distributed as dist

from .index import FlatIndex, merge_topk_packed_device, packed_result_block


def partition(n_total: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous row ranges [(lo, hi)) per rank; sizes differ by at most one row."""
    return [(n_total * r // world, n_total * (r + 1) // world) for r in range(world)]


class ShardedFlatIndex:
    def __init__(self, dim: int, n_total: int, rank: int | None = None, world: int | None = None,
                 device: int = 0, group=None, index_factory: Callable | None = None,
                 merge_fn: Callable | None = None):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.dim = dim
        self.n_total = n_total
        self.device = device
        self.lo, self.hi = partition(n_total, self.world)[self.rank]
        self.index = (index_factory or (lambda: FlatIndex(dim, key=None, device=device)))()
        self.index.reserve(self.hi - self.lo)
        self.index.set_id_offset(self.lo)
        self._merge = merge_fn  # None: native merge of the gathered blocks

    def owns(self, global_row: int) -> bool:
        return self.lo <= global_row < self.hi

    def add_local(self, rows) -> int:
        """Append this rank's rows (in global order).  Host array or device tensor."""
        if isinstance(rows, torch.Tensor) and rows.is_cuda:
            return self.index.add_device(rows)
        return self.index.add(rows.cpu().numpy() if isinstance(rows, torch.Tensor) else rows)

    def search(self, q: torch.Tensor, k: int):
        """q: [B, dim] tensor on this rank's device (same on every rank) ->
        (ids int64 [B,k], dists f32 [B,k], scores f32 [B,k]) identical on every rank."""
        B = q.shape[0]
        dev = q.device
        # results land in one block [ids | dists] so that ONE all-gather moves them (B*k*12 bytes per rank)
        block, ids, dists = packed_result_block(B, k, dev)
        scores = torch.zeros((B, k), dtype=torch.float32, device=dev)
        nf = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.index.search_device(q, k, ids, scores, dists, nf)
        if self.world == 1:
            return ids, dists, scores
        g_block = torch.zeros((self.world, block.numel()), dtype=torch.uint8, device=dev)
        # list-of-views form: accepted by both RCCL ("nccl") and gloo (CPU tests)
        dist.all_gather(list(g_block.unbind(0)), block, group=self.group)
        m_ids = torch.zeros((B, k), dtype=torch.int64, device=dev)
        m_dists = torch.zeros((B, k), dtype=torch.float32, device=dev)
        m_scores = torch.zeros((B, k), dtype=torch.float32, device=dev)
        if self._merge is None:
            # the merge is enqueued on torch's stream, behind the collective and the zero-fills above
            # (stream contract: include/memex_hip.h); the caller synchronises when it reads the results
            merge_topk_packed_device(self.device, g_block, self.world, B, k, m_ids, m_dists, m_scores,
                                     stream=torch.cuda.current_stream(dev).cuda_stream)
        else:  # injected merge (tests): takes the unpacked [G, B, k] arrays
            g_ids = g_block[:, : B * k * 8].contiguous().view(torch.int64).view(self.world, B, k)
            g_dists = g_block[:, B * k * 8:].contiguous().view(torch.float32).view(self.world, B, k)
            self._merge(g_ids, g_dists, m_ids, m_dists, m_scores)
        return m_ids, m_dists, m_scores

    def close(self) -> None:
        self.index.close()
