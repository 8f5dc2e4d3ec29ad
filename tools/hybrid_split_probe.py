#!/usr/bin/env python3
"""
How many workgroups does the dense scan need, and can BM25 run on the CUs it leaves?  config 2 shapes (1M chunks,
256 queries per call, depth 50): for every `spare` (CUs the scan does not take, hipidx_set_spare_cus) time the dense
leg alone, and the hybrid call with the dense leg launched first / BM25 launched first.  One JSON line per setting.
"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "intool-rag_amd"))


def main():
    import torch
    from hiprag import HipBM25, HipFlatIndex, build_postings, rrf_fuse_device
    dev = torch.device("cuda", 0)
    N, V, d, nq, depth, k = int(os.environ.get("DOCS", 1_000_000)), 262144, 1024, 256, 50, 10
    index = HipFlatIndex(d, "ip")
    for c in range(0, N, 125000):
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + c // 125000)
        x = torch.randn((min(125000, N - c), d), generator=g, device=dev)
        x /= x.norm(dim=1, keepdim=True)
        index.add_device(x)
    del x
    gq = torch.Generator(device=dev)
    gq.manual_seed(4321)
    queries = torch.randn((nq, d), generator=gq, device=dev)
    queries /= queries.norm(dim=1, keepdim=True)
    i = torch.arange(N, dtype=torch.int64, device=dev)
    doc_len = 64 + (i * 2654435761) % 256
    cdf = torch.cumsum(1.0 / torch.arange(1, V + 1, dtype=torch.float64, device=dev), 0)
    cdf /= cdf[-1].clone()
    gt = torch.Generator(device=dev)
    gt.manual_seed(777)
    u = torch.rand(int(doc_len.sum().item()), generator=gt, device=dev, dtype=torch.float64)
    term = torch.clamp(torch.searchsorted(cdf, u), max=V - 1)
    doc = torch.repeat_interleave(i, doc_len)
    postings = build_postings(doc.cpu().numpy(), term.cpu().numpy(), N, V, doc_len.cpu().numpy())
    del u, term, doc, i, cdf
    bm25 = HipBM25(postings, device=0)
    rng = np.random.default_rng(888)
    w = 1.0 / np.arange(17, V + 1, dtype=np.float64)
    cdfq = np.cumsum(w) / w.sum()
    sq = []
    for _ in range(nq):
        t = []
        while len(t) < 6:
            c = int(min(np.searchsorted(cdfq, rng.random()), len(cdfq) - 1)) + 16
            if c not in t:
                t.append(c)
        sq.append(np.asarray(t, dtype=np.uint32))
    side = torch.cuda.Stream(device=dev)
    main_s = torch.cuda.current_stream(dev)

    hp = torch.cuda.Stream(device=dev, priority=-1)
    from hiprag import hybrid_search_device

    def hybrid_hp():
        ready = torch.cuda.Event()
        ready.record(main_s)
        side.wait_event(ready)
        hp.wait_event(ready)
        with torch.cuda.stream(hp):
            dense = index.search_device(queries, depth)
            ddone = torch.cuda.Event()
            ddone.record(hp)
        with torch.cuda.stream(side):
            sparse = bm25.search_device(sq, depth)
            done = torch.cuda.Event()
            done.record(side)
        main_s.wait_event(ddone)
        main_s.wait_event(done)
        return rrf_fuse_device(dense[2], sparse[2], k)

    def hybrid_hp_main():      # dense on the high-priority stream, BM25 on the caller's stream
        ready = torch.cuda.Event()
        ready.record(main_s)
        hp.wait_event(ready)
        with torch.cuda.stream(hp):
            dense = index.search_device(queries, depth)
            ddone = torch.cuda.Event()
            ddone.record(hp)
        sparse = bm25.search_device(sq, depth)
        main_s.wait_event(ddone)
        return rrf_fuse_device(dense[2], sparse[2], k)

    def hybrid_bm25_then_hp():      # BM25 enqueued first on the caller's stream, then the dense leg on the high-priority stream
        ready = torch.cuda.Event()
        ready.record(main_s)
        hp.wait_event(ready)
        sparse = bm25.search_device(sq, depth)
        with torch.cuda.stream(hp):
            dense = index.search_device(queries, depth)
            ddone = torch.cuda.Event()
            ddone.record(hp)
        main_s.wait_event(ddone)
        return rrf_fuse_device(dense[2], sparse[2], k)

    def hybrid(dense_first):
        ready = torch.cuda.Event()
        ready.record(main_s)
        side.wait_event(ready)
        if dense_first:
            dense = index.search_device(queries, depth)
        with torch.cuda.stream(side):
            sparse = bm25.search_device(sq, depth)
            done = torch.cuda.Event()
            done.record(side)
        if not dense_first:
            dense = index.search_device(queries, depth)
        main_s.wait_event(done)
        sparse[2].record_stream(main_s)
        return rrf_fuse_device(dense[2], sparse[2], k)

    def timed(fn, steps=8, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    from collections import deque
    from hiprag.sharded import ShardedHybrid
    gq2 = torch.Generator(device=dev)
    gq2.manual_seed(999)
    many = torch.randn((2048, d), generator=gq2, device=dev)
    many /= many.norm(dim=1, keepdim=True)
    sq_many = [sq[i % nq] for i in range(2048)]

    def pipelined(sh, batch, steps=24):
        def run(n):
            pending = deque()
            last = None
            for i in range(n):
                o = (i * batch) % 2048
                pending.append(sh.search_begin(many[o:o + batch], sq_many[o:o + batch], depth, k))
                if len(pending) >= 4:
                    last = sh.search_end(pending.popleft())
            while pending:
                last = sh.search_end(pending.popleft())
            return last
        run(4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        return steps * batch / (time.perf_counter() - t0)

    ref = hybrid(False)
    torch.cuda.synchronize()
    ref = (ref[0].clone(), ref[1].clone())
    t_sparse = timed(lambda: bm25.search_device(sq, depth))
    print(json.dumps({"bm25_alone_qps": round(nq / t_sparse, 1)}), flush=True)
    spares = [int(s) for s in os.environ.get("SPARES", "0,32,64,80,96,112,128").split(",")]
    for spare in spares:
        index.set_spare_cus(spare)
        t_dense = timed(lambda: index.search_device(queries, depth))
        t_df = timed(lambda: hybrid(True))
        t_bf = timed(lambda: hybrid(False))
        t_hp = timed(hybrid_hp)
        t_hpm = timed(hybrid_hp_main)
        t_bhp = timed(hybrid_bm25_then_hp)
        t_prod = timed(lambda: hybrid_search_device(index, bm25, queries, sq, depth=depth, k=k))
        sh = ShardedHybrid(index, bm25)
        index.set_spare_cus(spare)
        pipe = {f"pipelined_{b}_per_step_qps": round(pipelined(sh, b), 1) for b in (256, 512)}
        print(json.dumps({"spare_cus": spare, **pipe}), flush=True)
        index.set_spare_cus(spare)
        got = hybrid(True)
        torch.cuda.synchronize()
        same = bool(torch.equal(got[1], ref[1]) and torch.equal(got[0], ref[0]))
        print(json.dumps({"spare_cus": spare, "dense_top50_qps": round(nq / t_dense, 1), "dense_ms": round(t_dense * 1e3, 3),
                          "hybrid_dense_first_qps": round(nq / t_df, 1), "hybrid_bm25_first_qps": round(nq / t_bf, 1),
                          "hybrid_dense_on_high_priority_stream_qps": round(nq / t_hp, 1), "product_call_qps": round(nq / t_prod, 1),
                          "hybrid_dense_high_priority_bm25_on_callers_stream_qps": round(nq / t_hpm, 1),
                          "hybrid_bm25_enqueued_first_dense_high_priority_qps": round(nq / t_bhp, 1),
                          "fused_equal_reference": same}), flush=True)


if __name__ == "__main__":
    main()
