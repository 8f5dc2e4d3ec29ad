"""
Row-sharded flat index: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), rank r holds
rows [lo_r, hi_r) of the corpus.  A search is: local exact top-k on every rank (global ids via id_base) -> ONE
all-gather of the packed partial lists ([2, nq, k] int64: fp64 score bits + ids, 16 B per entry) -> canonical
merge on every rank.  Because every row's score is the deterministic fp64 re-score of that row and the merge
comparator is (score, id), the sharded result equals the unsharded one bit for bit for any shard boundaries.

The reference is a single CPU process: this is new capability, not a port of anything.

Pipelining.  The index scan is the only HBM-heavy stage; selection, re-scoring, the exchange and the merge are
latency-bound.  `search_begin` therefore enqueues the scan on the caller's stream and everything after it on a
side stream (two workspace slots in the library), so that with one batch in flight the tail of batch i runs beside
the scan of batch i+1.  `search_end` makes the caller's stream wait for the batch's result.
"""
from __future__ import annotations

from typing import List, Tuple

from .index import HipFlatIndex, merge_topk_device

MAX_PASS = 32   # queries per scan pass (the N dimension of the MFMA tile)


def shard_bounds(n: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal row ranges; rank r gets [n*r//world, n*(r+1)//world)."""
    return [(n * r // world, n * (r + 1) // world) for r in range(world)]


def chunks_of_rank(n_chunks: int, world: int, rank: int) -> List[int]:
    """Contiguous chunk ids owned by `rank` when n_chunks generation chunks are dealt to `world` ranks in order."""
    return [c for c in range(n_chunks) if (c * world) // n_chunks == rank]


def all_gather_packed(pack, gathered, group=None, async_op: bool = True):
    """The ONE collective of a sharded search: every rank contributes `pack` ([2, nq, k] int64 = fp64 score bits and
    ids) and receives `gathered` ([world, 2, nq, k]).  RCCL ("nccl") gathers into the tensor directly; other backends
    (gloo in the CPU tests) take the list form over the same memory."""
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        return dist.all_gather_into_tensor(gathered, pack, group=group, async_op=async_op)
    return dist.all_gather([gathered[r] for r in range(gathered.shape[0])], pack, group=group, async_op=async_op)


class ShardedFlatIndex:
    def __init__(self, local: HipFlatIndex, row_lo: int = 0, group=None):
        import torch
        import torch.distributed as dist
        self.local = local
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        local.set_id_base(row_lo)
        # one side stream per workspace slot: the tail of batch i (finish -> all-gather -> merge) must not queue behind
        # the tail of batch i+1, which cannot start before scan i+1 ends
        self.side = [torch.cuda.Stream(device=local.device), torch.cuda.Stream(device=local.device)]
        self._slot = 0
        self._slot_done = [None, None]

    def search_begin(self, q, k: int):
        """q: float32 CUDA tensor [nq <= 32, d] that stays valid until search_end.  Returns a ticket."""
        import torch
        import torch.distributed as dist
        nq = q.shape[0]
        if nq > MAX_PASS:
            raise ValueError(f"search_begin takes at most {MAX_PASS} queries; use search_device for larger batches")
        slot, self._slot = self._slot, self._slot ^ 1
        main = torch.cuda.current_stream()
        if self._slot_done[slot] is not None:
            main.wait_event(self._slot_done[slot])      # the pass that last used this workspace slot has finished
        pack = torch.empty((2, nq, k), dtype=torch.int64, device=q.device)
        s32 = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        gathered = (torch.empty((self.world, 2, nq, k), dtype=torch.int64, device=q.device) if self.world > 1 else None)
        side = self.side[slot]
        for t in (pack, s32, gathered):
            if t is not None:
                t.record_stream(side)
        self.local.search_begin(q, k, slot)
        scanned = torch.cuda.Event()
        scanned.record(main)
        work = None
        with torch.cuda.stream(side):
            side.wait_event(scanned)
            self.local.search_finish(q, k, slot, (pack[0].view(torch.float64), s32, pack[1]))
            done = torch.cuda.Event()
            done.record(side)
            self._slot_done[slot] = done
            if self.world > 1:
                work = all_gather_packed(pack, gathered, self.group, async_op=True)
        return (work, pack, s32, gathered, k, slot, done)

    def search_end(self, ticket):
        """-> (scores64, scores32, ids) [nq,k]; the caller's current stream is made to wait for them."""
        import torch
        work, pack, s32, gathered, k, slot, done = ticket
        main = torch.cuda.current_stream()
        if work is None:
            main.wait_event(done)
            return (pack[0].view(torch.float64), s32, pack[1])
        side = self.side[slot]
        with torch.cuda.stream(side):
            work.wait()              # side stream waits for the collective; the host does not block
            out = merge_topk_device(gathered[:, 0].view(torch.float64), gathered[:, 1], k, self.local.metric)
            fin = torch.cuda.Event()
            fin.record(side)
        for t in out:
            t.record_stream(main)
        main.wait_event(fin)
        return out

    def search_device(self, q, k: int):
        """Any number of queries; passes of 32 are pipelined internally."""
        import torch
        nq = q.shape[0]
        if nq <= MAX_PASS:
            return self.search_end(self.search_begin(q, k))
        outs, ticket = [], None
        for o in range(0, nq, MAX_PASS):
            t = self.search_begin(q[o:o + MAX_PASS], k)
            if ticket is not None:
                outs.append(self.search_end(ticket))
            ticket = t
        outs.append(self.search_end(ticket))
        return tuple(torch.cat([o[i] for o in outs], dim=0) for i in range(3))


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
