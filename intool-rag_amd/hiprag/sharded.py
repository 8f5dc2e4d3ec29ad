"""
Row-sharded flat index: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), rank r holds
rows [lo_r, hi_r) of the corpus.  A search is: local exact top-k on every rank (global ids via id_base) -> ONE
all-gather of the packed partial lists ([2, nq, k] int64: fp64 score bits + ids, 16 B per entry) -> canonical
merge on every rank.  Because every row's score is the deterministic fp64 re-score of that row and the merge
comparator is (score, id), the sharded result equals the unsharded one bit for bit for any shard boundaries.

The reference is a single CPU process: this is new capability, not a port of anything.

The exchange is latency-bound (tens of KiB), so `search_begin` / `search_end` let the caller keep one batch in
flight: batch i's all-gather runs on RCCL's stream while batch i+1 scans.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

from .index import HipFlatIndex, merge_topk_device


def shard_bounds(n: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal row ranges; rank r gets [n*r//world, n*(r+1)//world)."""
    return [(n * r // world, n * (r + 1) // world) for r in range(world)]


class ShardedFlatIndex:
    def __init__(self, local: HipFlatIndex, row_lo: int, group=None):
        import torch.distributed as dist
        self.local = local
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        local.set_id_base(row_lo)

    def search_begin(self, q, k: int):
        """Enqueue the local search and the all-gather; returns a ticket for search_end."""
        import torch
        import torch.distributed as dist
        nq = q.shape[0]
        if self.world == 1:          # nothing to exchange: the local result is the global one
            return (None, self.local.search_device(q, k), k)
        pack = torch.empty((2, nq, k), dtype=torch.int64, device=q.device)
        s32 = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        self.local.search_device(q, k, out=(pack[0].view(torch.float64), s32, pack[1]))
        gathered = torch.empty((self.world, 2, nq, k), dtype=torch.int64, device=q.device)
        work = dist.all_gather_into_tensor(gathered, pack, group=self.group, async_op=True)
        return (work, gathered, k)

    def search_end(self, ticket, out=None):
        """Merge the gathered partial lists -> (scores64, scores32, ids) [nq,k] on every rank."""
        import torch
        work, gathered, k = ticket
        if work is None:
            return gathered
        work.wait()              # makes the current stream wait for the collective; the host does not block
        return merge_topk_device(gathered[:, 0].view(torch.float64), gathered[:, 1], k, self.local.metric, out)

    def search_device(self, q, k: int):
        return self.search_end(self.search_begin(q, k))


class EmulatedShards:
    """The same partition/merge logic with all shards on ONE device and no collective (tests, 1-GPU rehearsal)."""

    def __init__(self, shards: List[HipFlatIndex], bounds: List[Tuple[int, int]]):
        self.shards = shards
        for ix, (lo, _) in zip(shards, bounds):
            ix.set_id_base(lo)

    def search_device(self, q, k: int):
        import torch
        nq = q.shape[0]
        gathered = torch.empty((len(self.shards), 2, nq, k), dtype=torch.int64, device=q.device)
        s32 = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        for g, ix in enumerate(self.shards):
            ix.search_device(q, k, out=(gathered[g, 0].view(torch.float64), s32, gathered[g, 1]))
        return merge_topk_device(gathered[:, 0].view(torch.float64), gathered[:, 1], k, self.shards[0].metric)
