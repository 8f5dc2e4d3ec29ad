"""
Row-sharded flat index: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), rank r holds
rows [lo_r, hi_r) of the corpus.  A search is: local exact top-k on every rank (global ids via id_base) -> ONE
all-gather of the packed partial lists ([2, nq, k] int64: fp64 score bits + ids, 16 B per entry) -> canonical
merge on every rank.  Because every row's score is the deterministic fp64 re-score of that row and the merge
comparator is (score, id), the sharded result equals the unsharded one bit for bit for any shard boundaries.

The reference is a single CPU process: this is new capability, not a port of anything.

Pipelining.  The index scan is the only HBM-heavy stage; the finish, the exchange and the merge are latency-bound, and a
scan workgroup takes its CU's whole LDS, so nothing shares a CU with the scan.  But the scan does not need every CU: it is
HBM-bound, and a launch over 1M x 1024 rows takes the same time on 208 workgroups as on 256.  So `search_begin` enqueues the
scan (one launch of up to `launch_queries` queries = several passes over the local rows) on the index's own HIGH-PRIORITY
stream with SPARE_CUS CUs left out of its grid, and everything after it -- the finish, at N > 1 the all-gather, whose
ranks spin until every peer has joined, and the merge -- on the stream of one of eight workspace slots, where it runs
BESIDE the next scan on the CUs that scan leaves; `search_end` makes the caller's stream wait for the result.  The
priority is what makes the arrangement work: the next scan and the previous finish become ready at the same moment, and
with equal priorities the finish's many small workgroups are placed first, on every CU, and the scan's workgroups (which
need an EMPTY CU) start late and unevenly -- measured 2.68-2.87 ms per launch against 2.49 alone, which is why rounds 1-2
ran the finish in stream order on one GPU.  With the scan served first: 1M rows 194.7 -> 199.5 k queries/s, 500 k rows
374 -> 385 k, 125 k rows (the 8-GPU share) 1.09 -> 1.22 M, same box, A/B (`tools/scan_split_probe.py`); the idle time
between two scans drops from 0.10-0.15 ms to 0.024.  Priority alone still left the outcome to which hardware queues the
streams happened to get (0.89-1.23 M on the small shard), so the order is explicit: the tails of step i are enqueued when
scan i + 1 has been launched, behind a gate that scan opens once all its workgroups have started (hipidx_gate_tail_dev);
a step whose results are wanted before another scan is due gets its tails at search_end, ungated.  `tails_aside=False`
keeps the old arrangement (scan and finish in stream order on the caller's stream) for A/B runs.
"""
from __future__ import annotations

import os
from typing import List, Tuple

from .index import HipFlatIndex, merge_topk_device

N_SLOTS = 8     # library workspace slots = passes that may be in flight (DenseIndex::kSlots)
SPARE_CUS = 48          # CUs a flat scan leaves to the tails of earlier launches (of 256; 32-64 measure the same)
SPARE_CUS_HYBRID = 96   # ... when the tails include the BM25 leg (lib.cpp kHybridSpareCus)


def _scan_stream(device):
    """The device's scan stream (hiprag_scan_stream: the library's one high-priority stream per device, module docstring)
    as a torch stream."""
    import ctypes
    import torch
    from . import _native as nat
    st = _SCAN_STREAMS.get(device)
    if st is None:
        ptr = ctypes.c_void_p()
        nat.call("hiprag_scan_stream", int(device), ctypes.byref(ptr))
        st = _SCAN_STREAMS[device] = torch.cuda.ExternalStream(ptr.value, device=torch.device("cuda", int(device)))
    return st


def _order_behind(scan, main, ev) -> None:
    """Whatever `main` (the caller's stream) holds now is ahead of what is enqueued on `scan` next.  An idle caller's stream
    -- the steady state of a pipelined loop -- completes the marker within microseconds, and then the scan stream needs no
    barrier packet in front of the scan at all (one is ~10 us between two scans); a busy one gets the barrier."""
    ev.record(main)
    for _ in range(8):
        if ev.query():
            return
    scan.wait_event(ev)


_SCAN_STREAMS = {}


def shard_bounds(n: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, near-equal row ranges; rank r gets [n*r//world, n*(r+1)//world)."""
    return [(n * r // world, n * (r + 1) // world) for r in range(world)]


def chunks_of_rank(n_chunks: int, world: int, rank: int) -> List[int]:
    """Contiguous chunk ids owned by `rank` when n_chunks generation chunks are dealt to `world` ranks in order."""
    return [c for c in range(n_chunks) if (c * world) // n_chunks == rank]


def all_gather_packed(pack, gathered, group=None, async_op: bool = True):
    """The ONE collective of a sharded search: every rank contributes `pack` (int64: fp64 score bits and ids of its partial
    lists) and receives `gathered` ([world, *pack.shape]) -- one RCCL all_gather_into_tensor over xGMI.  Hosts and tests
    with another transport (the two-ranks-on-one-GPU rehearsals over gloo) pass their own callable with this signature as
    the `gather` argument of the sharded classes."""
    import torch.distributed as dist
    return dist.all_gather_into_tensor(gathered, pack, group=group, async_op=async_op)


def _slot_streams(device):
    """The streams the tails of the N_SLOTS workspace slots run on: the device's two tail streams (hiprag_tail_stream, owned by
    the library, ONE pair per device and process), the slots taking turns -- HIPRAG_SIDE_STREAMS=1 puts every slot on the
    first.  Two, so that the tail of batch i never queues behind that of batch i + 1 around an exchange.  One pair per
    PROCESS because HIP maps streams onto four hardware queues per priority in order of first use and lets later streams
    share queues: with a pair per ShardedFlatIndex object a process that built several of them (bench.py's legs) ran the later
    ones with shared queues, and streams that share a queue execute in submission order.  More hardware queues
    (GPU_MAX_HW_QUEUES = 8 / 16) or high-priority tail streams (a second queue pool) are time-sliced and lengthen the gap
    between launches two- to fivefold instead."""
    import ctypes
    import torch
    from . import _native as nat
    n = max(1, min(2, int(os.environ.get("HIPRAG_SIDE_STREAMS", "2"))))
    pool = []
    for which in range(n):
        st = _TAIL_STREAMS.get((device, which))
        if st is None:
            ptr = ctypes.c_void_p()
            nat.call("hiprag_tail_stream", int(device), which, ctypes.byref(ptr))
            st = _TAIL_STREAMS[(device, which)] = torch.cuda.ExternalStream(ptr.value, device=torch.device("cuda", int(device)))
        pool.append(st)
    return [pool[i % n] for i in range(N_SLOTS)]


_TAIL_STREAMS = {}


def _exchange_on(world: int) -> bool:
    import torch.distributed as dist
    return world > 1 or (dist.is_initialized() and os.environ.get("HIPRAG_FORCE_EXCHANGE") == "1")


def agree_min(value: int, device: int, group=None) -> int:
    """MIN of an integer over the ranks of `group` (one small all-reduce, at construction time only)."""
    import torch
    import torch.distributed as dist
    on_gpu = dist.get_backend(group) == "nccl"
    t = torch.tensor([int(value)], dtype=torch.int64, device=torch.device("cuda", device) if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())


def check_same_shape(shape: tuple, group=None) -> None:
    """Debug aid (HIPRAG_CHECK_SHAPES=1): every rank must enter a collective with the same (nq, k, ...)."""
    import torch.distributed as dist
    seen = [None] * dist.get_world_size(group)
    dist.all_gather_object(seen, tuple(int(v) for v in shape), group=group)
    if any(s != seen[0] for s in seen):
        raise RuntimeError(f"ranks disagree on the batch shape of a sharded search: {seen}")


class ShardedFlatIndex:
    def __init__(self, local: HipFlatIndex, row_lo: int = 0, group=None, gather=None, tails_aside=None):
        """tails_aside: None / True = the finish (and the exchange) run beside later scans, the scans on the high-priority
        stream (module docstring); False = scan and finish in stream order on the caller's stream (A/B measurements; not
        with an exchange)."""
        import torch
        import torch.distributed as dist
        self.local = local
        self.group = group
        self.gather = gather or all_gather_packed
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # the exchange (one all-gather per batch + merge) runs whenever there is more than one rank;
        # HIPRAG_FORCE_EXCHANGE=1 runs it on a one-rank group as well (a single-GPU box can then exercise the RCCL calls)
        self.exchange = _exchange_on(self.world)
        self.tails_aside = True if tails_aside is None else bool(tails_aside) or self.exchange
        local.set_id_base(row_lo)
        # CUs the scan never takes: the tails of earlier launches run there beside it -- the finish, and with an exchange the
        # all-gather kernel, which spins until every rank has launched it (a rank that reaches its collective early must not
        # hold CUs its own next scan is partitioned over)
        local.set_spare_cus(SPARE_CUS if self.tails_aside else 0)
        self.scan = _scan_stream(local.device) if self.tails_aside else None
        # the tails (finish -> all-gather -> merge) run on two streams that the slots take turns on (_slot_streams): the
        # tail of batch i never queues behind that of batch i+1, which cannot start before scan i+1 ends; its merge,
        # enqueued at search_end, may sit behind the finish of batch i+2 -- a scan that is ahead of it on the caller's
        # stream anyway
        self.side = _slot_streams(local.device)
        self._side_ptr = [st.cuda_stream for st in self.side]
        self._slot = 0
        self._pending = None          # slot of the step whose tails have not been enqueued yet (at most one: the newest)
        self._slot_used = [False] * N_SLOTS
        self._slot_ended = [False] * N_SLOTS
        self._bufs = [dict() for _ in range(N_SLOTS)]
        self._scan_done = [torch.cuda.Event() for _ in range(N_SLOTS)]
        self._max_pass = agree_min(local.launch_queries, local.device, group) if self.exchange else None
        self._check_shapes = self.exchange and os.environ.get("HIPRAG_CHECK_SHAPES") == "1"

    @property
    def max_pass(self) -> int:
        """Queries one search_begin takes.  One GPU: hipidx_launch_queries of the local index (sized by the shard: more
        passes per launch on a smaller shard, so a launch lasts about as long whatever the shard size).  Several ranks:
        the MINIMUM of that over the group, agreed once at construction -- shards may differ by a 32-row block, and near a
        rounding boundary of the sizing rule two ranks would otherwise cut the same batch into different numbers of
        all-gathers and hang (or exchange mismatched shapes)."""
        return self._max_pass if self._max_pass is not None else self.local.launch_queries

    def _buffers(self, slot: int, nq: int, k: int, dev):
        """Per-slot result buffers and events, created once per (nq, k): the steady state allocates nothing."""
        import torch
        key = (nq, k)
        cache = self._bufs[slot]
        if cache.get("key") != key:
            if self._slot_used[slot]:             # kernels of the slot's previous shape may still read the old buffers
                self.side[slot].synchronize()
                if not self.tails_aside:
                    torch.cuda.current_stream().synchronize()
            cache.clear()
            cache["key"] = key
            cache["pack"] = torch.empty((2, nq, k), dtype=torch.int64, device=dev)
            cache["s32"] = torch.empty((nq, k), dtype=torch.float32, device=dev)
            cache["gathered"] = (torch.empty((self.world, 2, nq, k), dtype=torch.int64, device=dev)
                                 if self.exchange else None)
            cache["merged"] = ((torch.empty((nq, k), dtype=torch.float64, device=dev),
                                torch.empty((nq, k), dtype=torch.float32, device=dev),
                                torch.empty((nq, k), dtype=torch.int64, device=dev)) if self.exchange else None)
            cache["scanned"], cache["done"], cache["fin"] = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
            cache["ready"] = torch.cuda.Event()
        return cache

    def search_begin(self, q, k: int):
        """q: float32 CUDA tensor [nq <= max_pass, d] that stays valid until search_end.  Returns a ticket.  The result
        tensors handed out by search_end belong to the slot and are reused N_SLOTS passes later."""
        import torch
        nq = q.shape[0]
        if nq > self.max_pass:
            raise ValueError(f"search_begin takes at most {self.max_pass} queries; use search_device for larger batches")
        if self._check_shapes:
            check_same_shape((nq, k), self.group)
        slot, self._slot = self._slot, (self._slot + 1) % N_SLOTS
        main = torch.cuda.current_stream()
        c = self._buffers(slot, nq, k, q.device)
        side = self.side[slot]
        pack = c["pack"]
        if not self.tails_aside:
            # A/B arrangement: scan and finish in stream order on the caller's stream (module docstring)
            self.local.search_begin(q, k, slot, stream=main.cuda_stream)
            self.local.search_finish(q, k, slot, (pack[0].view(torch.float64), c["s32"], pack[1]), stream=main.cuda_stream)
            self._slot_used[slot] = True
            self._slot_ended[slot] = True
            return (None, slot, k)
        # Scans are chained on the index's high-priority stream (one after the other, so a scan never shares the CUs with
        # another scan and HIP events around a launch measure that launch), behind whatever the caller's stream has
        # enqueued so far (the queries; the caller's reads of the results this slot held eight batches ago); everything
        # after the scan runs on the slot's own stream beside the next scans.
        scan = self.scan
        _order_behind(scan, main, c["ready"])
        if self._slot_used[slot] and not self._slot_ended[slot]:
            # the finish that last read this slot's workspace (its exchange and merge follow it on `side`); after a
            # search_end(wait=True) the caller's stream has waited for it already
            scan.wait_event(c["done"])
        self.local.search_begin(q, k, slot, stream=scan.cuda_stream)
        c["scanned"].record(scan)
        self._slot_used[slot] = True
        self._slot_ended[slot] = False
        # The tails of the PREVIOUS step are enqueued only now, behind a gate that the scan just launched opens when it has
        # STARTED (hipidx_gate_tail_dev): they are dispatched onto the CUs this scan leaves instead of racing it for CUs --
        # both become ready at the same moment, and which queue the hardware serves first depends on things no host controls.
        # This step's own tails wait for the next search_begin, or for its search_end, whichever comes first.
        self._flush_tails(gated=True)
        c["work"], c["q"] = None, q
        self._pending = slot
        return (None, slot, k)

    def _flush_tails(self, gated: bool) -> None:
        """Enqueue the tails (finish -> all-gather) of the step whose scan was launched last, if they are still due."""
        import torch
        slot = self._pending
        if slot is None:
            return
        self._pending = None
        c = self._bufs[slot]
        side, pack = self.side[slot], c["pack"]
        q, k = c["q"], c["key"][1]
        side.wait_event(c["scanned"])
        if gated:
            self.local.gate_tail(slot, self._side_ptr[slot])
        self.local.search_finish(q, k, slot, (pack[0].view(torch.float64), c["s32"], pack[1]), stream=self._side_ptr[slot])
        c["done"].record(side)
        if self.exchange:
            with torch.cuda.stream(side):
                c["work"] = self.gather(pack, c["gathered"], self.group, async_op=True)

    def search_end(self, ticket, wait: bool = True):
        """-> (scores64, scores32, ids) [nq,k].  With wait=True (default) the caller's current stream is made to wait
        for them; wait=False only finishes enqueueing (merge after the exchange) and leaves ordering to the caller --
        `result_event(ticket)` gives the event to wait on, a device synchronise covers everything."""
        import torch
        work, slot, k = ticket
        c = self._bufs[slot]
        main = torch.cuda.current_stream()
        if self.tails_aside:
            if self._pending == slot:
                self._flush_tails(gated=False)     # no scan has been launched behind this step's: nothing to wait for
            work = c["work"]
        self._slot_ended[slot] = wait or not self.tails_aside
        if work is None:
            if self.tails_aside:
                if wait:
                    main.wait_event(c["done"])
            # else: the results are in stream order on the stream search_begin was called on
            return (c["pack"][0].view(torch.float64), c["s32"], c["pack"][1])
        side = self.side[slot]
        with torch.cuda.stream(side):
            work.wait()              # side stream waits for the collective; the host does not block
            g = c["gathered"]
            out = merge_topk_device(g[:, 0].view(torch.float64), g[:, 1], k, self.local.metric, out=c["merged"])
            c["fin"].record(side)
        if wait:
            main.wait_event(c["fin"])
        return out

    def result_event(self, ticket):
        """Event that completes when the results of `ticket` are final (after search_end)."""
        work, slot, _ = ticket
        if self.tails_aside:
            work = self._bufs[slot].get("work")
        if work is None and not self.tails_aside:
            import torch
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            return ev
        return self._bufs[slot]["fin" if work is not None else "done"]

    def search_device(self, q, k: int):
        """Any number of queries; passes of max_pass queries are pipelined internally."""
        import torch
        nq = q.shape[0]
        if nq <= self.max_pass:
            return self.search_end(self.search_begin(q, k))
        out = (torch.empty((nq, k), dtype=torch.float64, device=q.device),
               torch.empty((nq, k), dtype=torch.float32, device=q.device),
               torch.empty((nq, k), dtype=torch.int64, device=q.device))
        pending = []

        def drain():
            o, t = pending.pop(0)
            res = self.search_end(t)          # slot-owned buffers: copy out before the slot is reused
            for dst, src in zip(out, res):
                dst[o:o + src.shape[0]].copy_(src)

        for o in range(0, nq, self.max_pass):
            pending.append((o, self.search_begin(q[o:o + self.max_pass], k)))
            if len(pending) >= N_SLOTS - 1:
                drain()
        while pending:
            drain()
        return out


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


# ---------------------------------------------------------------------------------------------------------
# Row-sharded HYBRID search (BASELINE configs[3]): dense rows and BM25 postings are partitioned by the SAME
# document ranges (idf and avgdl are global constants baked into the impacts when the postings are built, so a shard's
# scores equal the unsharded ones).  A batch is: local dense top-`depth` + local BM25 top-`depth` -> ONE all-gather of
# both partial lists packed into a single buffer ([leg, {score bits, ids}, nq, depth] int64) -> every rank merges each
# leg with the canonical comparator (score, id) -> RRF over the two GLOBAL lists (ranks are global: fusing per shard
# and merging afterwards would be a different function).
# ---------------------------------------------------------------------------------------------------------
def _pack_hybrid(dense: HipFlatIndex, bm25, q, sparse_queries, depth: int, pack):
    import torch
    nq = q.shape[0]
    s32 = torch.empty((nq, depth), dtype=torch.float32, device=q.device)
    dense.search_device(q, depth, out=(pack[0, 0].view(torch.float64), s32, pack[0, 1]))
    bm25.search_device(sparse_queries, depth, out=(pack[1, 0].view(torch.float64), s32, pack[1, 1]))


def _merge_and_fuse(gathered, depth: int, k: int, metric, c: float, w_dense: float, w_sparse: float):
    """gathered: [parts, 2 legs, 2, nq, depth] int64 -> (fused scores float32 [nq,k], ids int64 [nq,k])."""
    import torch
    from ._native import METRIC_IP
    from .fusion import rrf_fuse_device
    dl = merge_topk_device(gathered[:, 0, 0].view(torch.float64), gathered[:, 0, 1], depth, metric)
    sl = merge_topk_device(gathered[:, 1, 0].view(torch.float64), gathered[:, 1, 1], depth, METRIC_IP)   # BM25: higher is better
    return rrf_fuse_device(dl[2], sl[2], k, c, w_dense, w_sparse)


class ShardedHybrid:
    """Row-sharded hybrid search with the same pipeline as ShardedFlatIndex, as a client of the C-ABI's two halves
    (hiphybrid_shard_begin_dev / hiphybrid_shard_end_dev, include/hiprag.h): `search_begin` = one library call -- the dense
    scan of the batch on the high-priority scan stream; its finish and the BM25 leg on the stream of one of eight slots, filling the
    two halves of ONE packed buffer -- then the ONE all-gather (async) on that stream; `search_end` = one library call:
    the two global merges and the fusion.  Pre-allocated per-slot buffers; batches larger than the agreed launch size are
    cut into pipelined pieces."""

    def __init__(self, dense: HipFlatIndex, bm25, row_lo: int = 0, group=None, gather=None):
        import torch.distributed as dist
        self.dense, self.bm25, self.group = dense, bm25, group
        self.gather = gather or all_gather_packed
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.exchange = _exchange_on(self.world)     # see ShardedFlatIndex
        dense.set_id_base(row_lo)
        bm25.set_id_base(row_lo)
        # the tails of a batch -- dense finish, BM25 leg, all-gather, merges, fusion -- run beside the next scans on the CUs
        # the scan grid leaves, the scans on the high-priority stream (ShardedFlatIndex, module docstring)
        dense.set_spare_cus(SPARE_CUS_HYBRID)
        self.scan = _scan_stream(dense.device)
        self._max_pass = agree_min(dense.launch_queries, dense.device, group) if self.exchange else None
        self._check_shapes = self.exchange and os.environ.get("HIPRAG_CHECK_SHAPES") == "1"
        self.side = _slot_streams(dense.device)
        self._side_ptr = [st.cuda_stream for st in self.side]
        self._slot = 0
        self._used = [False] * N_SLOTS
        self._ended = [True] * N_SLOTS
        self._bufs = [dict() for _ in range(N_SLOTS)]

    @property
    def max_pass(self) -> int:
        return self._max_pass if self._max_pass is not None else self.dense.launch_queries

    def _buffers(self, slot: int, nq: int, depth: int, k: int, dev):
        import torch
        key = (nq, depth, k)
        c = self._bufs[slot]
        if c.get("key") != key:
            if self._used[slot]:
                self.side[slot].synchronize()
            c.clear()
            c["key"] = key
            c["pack"] = torch.empty((2, 2, nq, depth), dtype=torch.int64, device=dev)       # [leg, {score bits, ids}, nq, depth]
            c["s32"] = torch.empty((2, nq, depth), dtype=torch.float32, device=dev)
            c["gathered"] = (torch.empty((self.world, 2, 2, nq, depth), dtype=torch.int64, device=dev)
                             if self.exchange else None)
            c["merged"] = torch.empty((4, nq, depth), dtype=torch.int64, device=dev)       # scratch of hiphybrid_shard_end_dev
            c["fused"] = (torch.empty((nq, k), dtype=torch.float32, device=dev),
                          torch.empty((nq, k), dtype=torch.int64, device=dev))
            c["fin"], c["ready"] = torch.cuda.Event(), torch.cuda.Event()
        return c

    def search_begin(self, q, sparse_queries, depth: int = 50, k: int = 10, c: float = 60.0, w_dense: float = 1.0,
                     w_sparse: float = 1.0):
        """q: float32 CUDA tensor [nq <= max_pass, d], the same on every rank, valid until search_end; sparse_queries: nq
        term-id lists.  Returns a ticket; the tensors search_end hands out belong to the slot (reused 8 batches later)."""
        import ctypes
        import torch
        from . import _native as nat
        nq = q.shape[0]
        if nq > self.max_pass or len(sparse_queries) != nq:
            raise ValueError(f"search_begin takes at most {self.max_pass} queries with one term list each")
        if self._check_shapes:
            check_same_shape((nq, depth, k), self.group)
        slot, self._slot = self._slot, (self._slot + 1) % N_SLOTS
        main = torch.cuda.current_stream()
        b = self._buffers(slot, nq, depth, k, q.device)
        side, pack = self.side[slot], b["pack"]
        scan = self.scan
        _order_behind(scan, main, b["ready"])    # the queries; the caller's reads of the results this slot held eight batches ago
        if self._used[slot] and not self._ended[slot]:
            scan.wait_event(b["fin"])            # the batch that last used the slot's workspace and buffers
        terms, qoff = self.bm25._flatten(sparse_queries)
        nat.call("hiphybrid_shard_begin_dev", self.dense._h, self.bm25._h, q.data_ptr(), terms.ctypes.data if terms.size else None,
                 qoff.ctypes.data, nq, int(depth), slot, pack.data_ptr(), b["s32"].data_ptr(), ctypes.c_void_p(scan.cuda_stream),
                 ctypes.c_void_p(self._side_ptr[slot]))
        self._used[slot], self._ended[slot] = True, False
        work = None
        if self.exchange:
            with torch.cuda.stream(side):
                work = self.gather(pack, b["gathered"], self.group, async_op=True)
        return (work, slot, depth, k, c, w_dense, w_sparse)

    def search_end(self, ticket, wait: bool = True):
        """-> (fused scores float32 [nq,k], ids int64 [nq,k]), identical on every rank."""
        import ctypes
        import torch
        from . import _native as nat
        work, slot, depth, k, c, w_dense, w_sparse = ticket
        b = self._bufs[slot]
        side = self.side[slot]
        nq = b["pack"].shape[2]
        with torch.cuda.stream(side):
            if work is not None:
                work.wait()
            g = b["gathered"] if self.exchange else b["pack"]
            nat.call("hiphybrid_shard_end_dev", g.data_ptr(), self.world if self.exchange else 1, nq, int(depth), int(k),
                     int(self.dense.metric), float(c), float(w_dense), float(w_sparse), b["merged"].data_ptr(),
                     b["fused"][0].data_ptr(), b["fused"][1].data_ptr(), ctypes.c_void_p(self._side_ptr[slot]))
            b["fin"].record(side)
        self._ended[slot] = wait
        if wait:
            torch.cuda.current_stream().wait_event(b["fin"])
        return b["fused"]

    def search_device(self, q, sparse_queries, depth: int = 50, k: int = 10, c: float = 60.0, w_dense: float = 1.0,
                      w_sparse: float = 1.0):
        """Any number of queries (the same on every rank); every rank returns the same fused (scores float32 [nq,k],
        ids int64 [nq,k])."""
        import torch
        nq = q.shape[0]
        if nq <= self.max_pass:
            return self.search_end(self.search_begin(q, sparse_queries, depth, k, c, w_dense, w_sparse))
        out = (torch.empty((nq, k), dtype=torch.float32, device=q.device),
               torch.empty((nq, k), dtype=torch.int64, device=q.device))
        pending = []

        def drain():
            o, t = pending.pop(0)
            for dst, src in zip(out, self.search_end(t)):
                dst[o:o + src.shape[0]].copy_(src)

        for o in range(0, nq, self.max_pass):
            pending.append((o, self.search_begin(q[o:o + self.max_pass], sparse_queries[o:o + self.max_pass], depth, k, c,
                                                 w_dense, w_sparse)))
            if len(pending) >= N_SLOTS - 1:
                drain()
        while pending:
            drain()
        return out


class EmulatedHybridShards:
    """ShardedHybrid's partition / merge / fuse with every shard on ONE device and no collective (tests)."""

    def __init__(self, shards, bounds: List[Tuple[int, int]]):
        self.shards = shards           # [(HipFlatIndex, HipBM25)], document ranges `bounds`
        for (ix, bm), (lo, _) in zip(shards, bounds):
            ix.set_id_base(lo)
            bm.set_id_base(lo)

    def search_device(self, q, sparse_queries, depth: int = 50, k: int = 10, c: float = 60.0, w_dense: float = 1.0,
                      w_sparse: float = 1.0):
        import torch
        nq = q.shape[0]
        gathered = torch.empty((len(self.shards), 2, 2, nq, depth), dtype=torch.int64, device=q.device)
        for g, (ix, bm) in enumerate(self.shards):
            _pack_hybrid(ix, bm, q, sparse_queries, depth, gathered[g])
        return _merge_and_fuse(gathered, depth, k, self.shards[0][0].metric, c, w_dense, w_sparse)
