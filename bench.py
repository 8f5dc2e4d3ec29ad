#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MI355X hybrid-retrieval hot path.

Metric (BASELINE.json): queries/sec (+ p50 retrieve latency) for exact top-10 inner-product search on a
1M x 1024-d fp32 index.  Workload = BASELINE configs[1]: "1M x 1024-d synthetic vectors, brute-force
inner-product top-10".  A STEP is one pass of the hot path over one batch of synthetic queries already resident in
HBM: ONE scan launch (8 passes of 64 queries at 1M rows; each pass streams the bf16 filter copy of the local rows
once, HBM-bound, and keeps a candidate list per query) -> ONE finish kernel: rank the list, fp64 re-score of the fp32 rows,
certificate (-> exhaustive path where it fails) -> (N>1: one RCCL all-gather of the packed partial top-k + canonical
merge).  Results are exact in every case.

Multi-GPU (one rank per GPU: the driver launches them through torch.distributed.run, and `python bench.py --gpus N` run bare
starts them the same way itself; a world size other than --gpus is fatal): the 1M rows are sharded row-wise
across the N ranks (STRONG scaling, total work fixed) and merged by one all-gather per step; up to four steps are
in flight so the latency-bound tail of a step (selection, fp64 re-score, exchange, merge) runs beside later scans.
The library sizes a launch by the local row count (16 passes = 1024 queries per step at <= 250k rows per GPU).

At N = 1 the same JSON line also carries `legs` (rank 0, after the headline section, each on the same GPU with its inputs
resident in HBM; `--legs none` skips them, N > 1 never runs them):
  legs.fp32_stream  the SURVEY 8(d)-priced scan: HIPRAG_SCAN_MODE=q64 streams the fp32 rows (N*d*4 bytes per pass) instead of
                    the bf16 filter copy -- same exact results, the number the survey's byte accounting refers to
  legs.reference_shape  the call shape the reference itself uses: L2 metric (faiss.IndexFlatL2, rag/storage/faiss_index.py:123), k = 50
                    (page_retriever.py:81), ONE query per call (:81-83) -- p50 / p99 through hipidx_search with host arrays and through
                    the overlay's search_hip_by_vector (enrichment included) -- plus batched L2 / k = 50 throughput and its roofline
  legs.ivf          IVF-Flat over the same rows (north_star's "flat-IP / IVF distance scan"): recall against the exact flat top-10, p50 of
                    one query per call and the bytes a query reads, at nprobe 1 / 8 / 32 of 1024 lists
  legs.hybrid       BASELINE configs[2]: 1M chunks, dense top-50 + BM25 term-at-a-time top-50 + RRF -> top-10; the BM25
                    roofline is priced on the posting bytes the queries REQUEST (sum df * 8 B), not on SURVEY 8(d)'s
                    accumulator passes, which the tiled kernel does not make
  legs.encoder      BASELINE configs[4]'s embed stage: 256 x 512 tokens through the 24-layer XLM-R-large-shaped encoder
                    (seeded random weights), flops / time / 2.5 PFLOP/s

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     dominant kernel = scan_bf16_kernel; achieved = algorithmic bytes per launch (passes * rows_local * d_pad * 2:
               the scan streams a 2-byte-per-element filter copy, the 4-byte rows are only touched by re-scoring;
               DESIGN.md) / mean launch duration from HIP events recorded on the launch stream around every scan
               launch of the timed region (in-kernel wall-clock stamps of the same launches beside them); peak 8000 GB/s (MI355X_MICROARCH.md).
  cpu_baseline the oracle's reference-faithful fp32 twin (one query per call, one thread) timed on this box's host
               cores on a bounded sample, rank 0, N=1 only.  A reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "intool-rag_amd"))

N_ROWS = 1_000_000
DIM = 1024
TOPK = 10
N_QUERIES = 4096
CHUNK = 31250                 # generation unit: N_ROWS/32, so shards of 1,2,4,8,16,32 ranks align to chunks
HBM_PEAK_GBS = 8000.0


def gen_chunk(torch, chunk_id: int, rows: int, dev):
    """Unit-norm Gaussian rows; content depends only on the chunk id, never on the sharding."""
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + chunk_id)
    x = torch.randn((rows, DIM), generator=g, device=dev, dtype=torch.float32)
    x /= x.norm(dim=1, keepdim=True)
    return x


def _time_steps(torch, fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def run_legs(torch, args, dev, index, queries, legs):
    """The single-GPU legs beside the headline line (module docstring).  Each leg frees what it allocated."""
    from hiprag import HipBM25, HipFlatIndex, build_postings, rrf_fuse_device
    from hiprag.sharded import ShardedFlatIndex
    out = {}
    n_rows = args.rows
    chunk = CHUNK if n_rows == N_ROWS else max(1, n_rows // 32)
    def leg_fp32():
        # SURVEY 8(d)'s accounting (N*d*4 bytes per pass): the scan that streams the fp32 rows themselves
        old = os.environ.get("HIPRAG_SCAN_MODE")
        os.environ["HIPRAG_SCAN_MODE"] = "q64"
        ix = HipFlatIndex(DIM, "ip", device=dev.index)
        if old is None:
            del os.environ["HIPRAG_SCAN_MODE"]
        else:
            os.environ["HIPRAG_SCAN_MODE"] = old
        for c in range((n_rows + chunk - 1) // chunk):
            ix.add_device(gen_chunk(torch, c, min(chunk, n_rows - c * chunk), dev))
        sh = ShardedFlatIndex(ix, 0)
        ix.reserve_search(TOPK)
        batch = ix.launch_queries
        passes = (batch + ix.pass_queries - 1) // ix.pass_queries
        nb = N_QUERIES // batch
        from collections import deque

        def steps(n, first=0):
            pending = deque()
            for s in range(n):
                b = (first + s) % nb
                pending.append(sh.search_begin(queries[b * batch:(b + 1) * batch], TOPK))
                if len(pending) >= 4:
                    sh.search_end(pending.popleft())
            while pending:
                sh.search_end(pending.popleft())
        steps(10)
        torch.cuda.synchronize()
        ix.enable_timing(1)
        torch.cuda.synchronize()
        n = max(20, args.steps // 4)
        t0 = time.perf_counter()
        steps(n, 10)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        st = ix.stats()
        ix.enable_timing(False)
        same = bool(torch.equal(ix.search_device(queries[:64], TOPK)[2], index.search_device(queries[:64], TOPK)[2]))
        bpl = int(st["bytes_per_pass"]) * passes
        ach = bpl / (st["avg_scan_ms"] * 1e-3) / 1e9
        out["fp32_stream"] = {"workload": "configs[1] with HIPRAG_SCAN_MODE=q64: the scan streams the fp32 rows (N*d*4 B per pass)",
                              "value": round(n * batch / el, 1), "unit": "queries/s", "steps": n, "queries_per_step": batch,
                              "ms_per_step": round(el / n * 1e3, 4), "ids_equal_default_mode": same,
                              "roofline": {"bound": "hbm", "kernel": "scan_split_kernel", "achieved": round(ach, 1),
                                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                           "bytes_per_launch": bpl, "passes_per_launch": passes,
                                           "avg_launch_ms": round(float(st["avg_scan_ms"]), 5)}}
        del sh, ix
        torch.cuda.empty_cache()

    def leg_reference():
        """The call shape the reference itself uses: faiss.IndexFlatL2 (rag/storage/faiss_index.py:123),
        ONE query per search call (:81-83), k = 50 (rag/query/page_retriever.py:81) -- through the C-ABI with host arrays and
        through the drop-in overlay (search_hip_by_vector: list in, enriched dicts out); plus batched L2 / k = 50 throughput
        with its roofline (the norms stream is part of the bytes)."""
        import asyncio
        import tempfile
        K = 50
        ix = HipFlatIndex(DIM, "l2", device=dev.index)
        ix.reserve_rows(n_rows)
        for c in range((n_rows + chunk - 1) // chunk):
            ix.add_device(gen_chunk(torch, c, min(chunk, n_rows - c * chunk), dev))
        ix.reserve_search(K)
        nwarm, nq1 = 20, 200
        q_host = queries[:nwarm + nq1].cpu().numpy()
        lat = []
        for i in range(nwarm + nq1):                       # (a) hipidx_search: host query in, host results out, one per call
            t1 = time.perf_counter()
            ix.search(q_host[i:i + 1], K)
            lat.append((time.perf_counter() - t1) * 1e3)
        lat = np.sort(np.asarray(lat[nwarm:]))
        # (b) the overlay: the resident index stands where the reference's path-keyed _INDEX_CACHE (faiss_index.py:24) would
        # hold the loaded file; the chunk table is a real {doc}_chunks.json of n_rows chunks, parsed once and cached
        tmp = tempfile.mkdtemp(prefix="hiprag_bench_")
        old_storage = os.environ.get("STORAGE_DIR")
        os.environ["STORAGE_DIR"] = tmp
        import rag.storage.hip_index as hi
        t1 = time.perf_counter()
        with open(os.path.join(tmp, "bench_chunks.json"), "w") as f:
            json.dump({"chunks": [{"chunk_id": f"c{i}", "text": f"chunk {i}", "page": 1 + i // 40,
                                   "metadata": {"title": f"T{i // 40}", "source_filename": "bench.pdf"}} for i in range(n_rows)]}, f)
        write_s = time.perf_counter() - t1
        ipath = os.path.join(tmp, "bench" + hi.INDEX_SUFFIX)
        open(ipath, "wb").close()
        hi.clear_caches()
        hi._INDEX_CACHE[ipath] = ix
        q_lists = [q_host[i].tolist() for i in range(nwarm + nq1)]
        t1 = time.perf_counter()
        first = asyncio.run(hi.search_hip_by_vector(q_lists[0], limit=K))          # parses and caches the chunk table
        first_s = time.perf_counter() - t1
        lat_o = []
        for i in range(nwarm + nq1):
            t1 = time.perf_counter()
            rows = asyncio.run(hi.search_hip_by_vector(q_lists[i], limit=K))
            lat_o.append((time.perf_counter() - t1) * 1e3)
        lat_o = np.sort(np.asarray(lat_o[nwarm:]))
        ids_ok = [int(r["chunk_id"][1:]) for r in rows] == [int(v) for v in ix.search(q_host[nwarm + nq1 - 1:nwarm + nq1], K)[1][0]]
        hi.clear_caches()
        if old_storage is None:
            del os.environ["STORAGE_DIR"]
        else:
            os.environ["STORAGE_DIR"] = old_storage
        # (c) batched: the pipelined launches of the headline section, L2 metric and k = 50
        sh = ShardedFlatIndex(ix, 0)
        batch = ix.launch_queries
        passes = (batch + ix.pass_queries - 1) // ix.pass_queries
        nb = N_QUERIES // batch
        from collections import deque

        def steps(n, first_=0):
            pending = deque()
            for s_ in range(n):
                b = (first_ + s_) % nb
                pending.append(sh.search_begin(queries[b * batch:(b + 1) * batch], K))
                if len(pending) >= 4:
                    sh.search_end(pending.popleft())
            while pending:
                sh.search_end(pending.popleft())
        steps(10)
        torch.cuda.synchronize()
        st0 = ix.stats()
        ix.enable_timing(1)
        torch.cuda.synchronize()
        n = max(20, args.steps // 4)
        t0 = time.perf_counter()
        steps(n, 10)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        st = ix.stats()
        ix.enable_timing(False)
        bpl = int(st["bytes_per_pass"]) * passes
        ach = bpl / (st["avg_scan_ms"] * 1e-3) / 1e9
        nqs = max(1, int(st["queries"] - st0["queries"]))
        out["reference_shape"] = {
            "workload": f"the reference's call shape on {n_rows} x {DIM}: L2 metric (IndexFlatL2), k = {K}, one query per call",
            "hipidx_search_p50_ms": round(float(lat[len(lat) // 2]), 4), "hipidx_search_p99_ms": round(float(lat[int(len(lat) * 0.99) - 1]), 4),
            "overlay_search_hip_by_vector_p50_ms": round(float(lat_o[len(lat_o) // 2]), 4),
            "overlay_search_hip_by_vector_p99_ms": round(float(lat_o[int(len(lat_o) * 0.99) - 1]), 4),
            "overlay_rows_returned": len(first), "overlay_ids_equal_hipidx_search": bool(ids_ok),
            "chunk_table_first_load_s": round(first_s, 2), "chunk_table_write_s": round(write_s, 2), "queries_timed": nq1,
            "batched": {"value": round(n * batch / el, 1), "unit": "queries/s", "k": K, "metric": "l2", "queries_per_step": batch,
                        "ms_per_step": round(el / n * 1e3, 4),
                        "extended_queries_share": round((st["roundb_queries"] - st0["roundb_queries"]) / nqs, 4),
                        "fallback_queries": int(st["fallback_queries"] - st0["fallback_queries"]),
                        "rescored_groups_per_query": round((st["rescored_groups"] - st0["rescored_groups"]) / nqs, 1),
                        "roofline": {"bound": "hbm", "kernel": "scan_bf16_kernel<L2>", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                                     "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "bytes_per_launch": bpl,
                                     "bytes": "bf16 filter copy + the fp32 squared norms of every row, per pass",
                                     "passes_per_launch": passes, "avg_launch_ms": round(float(st["avg_scan_ms"]), 5)}}}
        del sh, ix
        torch.cuda.empty_cache()

    def leg_ivf():
        """BASELINE north_star's other scan: IVF-Flat over the same 1M rows (nlist = 1024) -- the approximate one-query mode.
        Recall against the exact flat top-10 of the headline index, p50 of one query per call, and the bytes a query reads."""
        from hiprag import HipIVFIndex
        nlist = 1024
        t1 = time.perf_counter()
        x = torch.cat([gen_chunk(torch, c, min(chunk, n_rows - c * chunk), dev) for c in range((n_rows + chunk - 1) // chunk)])
        iv = HipIVFIndex(DIM, nlist, "ip", device=dev.index)
        iv.train_add(x, iters=6)
        del x
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t1
        nqr = 256
        truth = index.search_device(queries[:nqr], TOPK)[2]
        # SURVEY 8(d)'s "planted" queries: row i + 5 % noise, renormalised -- their nearest neighbour is known, and it lives in
        # ONE list; the isotropic random queries above have no such structure (their neighbours are spread evenly over the
        # lists, so recall on them is nprobe / nlist by construction: what IVF does on data without clusters)
        gp = torch.Generator(device=dev)
        gp.manual_seed(99)
        prow = torch.arange(nqr, device=dev) * 3907 + 11
        src = torch.cat([gen_chunk(torch, int(r) // chunk, min(chunk, n_rows - (int(r) // chunk) * chunk), dev)[int(r) % chunk][None] for r in prow.tolist()])
        noise = torch.randn(src.shape, generator=gp, device=dev)
        planted = src + 0.05 * noise / noise.norm(dim=1, keepdim=True)
        planted = (planted / planted.norm(dim=1, keepdim=True)).contiguous()
        torch.cuda.synchronize()
        lens = np.sort(iv.list_lengths)[::-1]
        res = {}
        for nprobe in (1, 8, 32):
            got = iv.search_device(queries[:nqr], TOPK, nprobe)[2]
            top1 = iv.search_device(planted, 1, nprobe)[2][:, 0]
            torch.cuda.synchronize()
            planted_hit = float((top1 == prow).float().mean().item())
            recall = float(np.mean([len(set(a.tolist()) & set(b.tolist())) / TOPK for a, b in zip(got.cpu().numpy(), truth.cpu().numpy())]))
            lat = []
            for i in range(120):
                qi = queries[i:i + 1]
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                iv.search_device(qi, TOPK, nprobe)
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t2) * 1e3)
            lat = np.sort(np.asarray(lat[20:]))
            tb = _time_steps(torch, lambda: iv.search_device(queries[:nqr], TOPK, nprobe), 5, 2)
            rows_q = nprobe * n_rows / nlist
            res[f"nprobe_{nprobe}"] = {"planted_neighbour_found": round(planted_hit, 4), "recall_at_10_isotropic_queries": round(recall, 4), "p50_ms_single_query": round(float(lat[len(lat) // 2]), 4),
                                       "batched_256_qps": round(nqr / tb, 1), "rows_scored_per_query_avg": int(rows_q),
                                       "bytes_per_query_avg": int(rows_q * DIM * 4),
                                       "batched_GBs": round(nqr * rows_q * DIM * 4 / tb / 1e9, 1)}
        out["ivf"] = {"workload": f"IVF-Flat over the same {n_rows} x {DIM} rows, nlist {nlist}, inner product, top-{TOPK}; exact fp64 scores "
                                  f"of the probed rows, approximate only in WHICH rows are probed (exact at nprobe = nlist)",
                      "build_s": round(build_s, 1), "longest_list": int(lens[0]), "shortest_list": int(lens[-1]), **res,
                      "flat_p50_ms_single_query_for_comparison": "see p50_ms_single_query_device_resident of the headline"}
        iv.close()
        torch.cuda.empty_cache()

    def leg_hybrid():
        N, V, depth, nq = n_rows, 262144, 50, 256
        t0 = time.time()
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
        build_s = time.time() - t0
        bm25 = HipBM25(postings, device=dev.index)
        rng = np.random.default_rng(888)
        w = 1.0 / np.arange(17, V + 1, dtype=np.float64)           # 6 distinct terms per query, Zipf restricted to ranks >= 16
        cdfq = np.cumsum(w) / w.sum()
        sq = []
        for _ in range(nq):
            t = []
            while len(t) < 6:
                c = int(min(np.searchsorted(cdfq, rng.random()), len(cdfq) - 1)) + 16
                if c not in t:
                    t.append(c)
            sq.append(np.asarray(t, dtype=np.uint32))
        qd = queries[:nq]
        b0 = bm25.stats()
        t_sparse = _time_steps(torch, lambda: bm25.search_device(sq, depth), 5, 2)
        b1 = bm25.stats()
        d0 = index.stats()
        t_dense = _time_steps(torch, lambda: index.search_device(qd, depth), 5, 2)
        d1 = index.stats()
        dq = max(1, int(d1["queries"] - d0["queries"]))
        dense_stats = {"extended_queries_share": round((d1["roundb_queries"] - d0["roundb_queries"]) / dq, 4),
                       "fallback_queries": int(d1["fallback_queries"] - d0["fallback_queries"]),
                       "list_entries_per_query": round((d1["list_entries"] - d0["list_entries"]) / dq, 1),
                       "ranked_entries_per_query": round((d1["ranked_entries"] - d0["ranked_entries"]) / dq, 1),
                       "rescored_groups_per_query": round((d1["rescored_groups"] - d0["rescored_groups"]) / dq, 1)}
        dl = index.search_device(qd, depth)
        sl = bm25.search_device(sq, depth)
        t_fuse = _time_steps(torch, lambda: rrf_fuse_device(dl[2], sl[2], TOPK), 20, 3)

        from hiprag import hybrid_search_device

        def hybrid():   # the product's hybrid call: BM25 on a helper stream beside the dense leg, RRF behind both
            return hybrid_search_device(index, bm25, qd, sq, depth=depth, k=TOPK)
        t_all = _time_steps(torch, hybrid, 5, 2)
        fs, fi = hybrid()
        ss, si = rrf_fuse_device(dl[2], sl[2], TOPK)
        torch.cuda.synchronize()
        same_fused = bool(torch.equal(fi, si) and torch.equal(fs, ss))
        calls = 7
        post_per_q = (b1["postings_touched"] - b0["postings_touched"]) / (calls * nq)
        posting_gbs = post_per_q * 8 * (nq / t_sparse) / 1e9
        fetch = None
        prof = os.path.join(REPO, "profiles", "r03_pmc_hybrid_fetch.json")
        if os.path.exists(prof):
            fetch = {"source": "profiles/r03_pmc_hybrid_fetch.json (rocprofv3 --pmc FETCH_SIZE of THIS round's kernels; not collected "
                               "by this run)", "summary": json.load(open(prof)).get("kernels", {}).get("taat_tile_kernel")}
        out["hybrid"] = {"workload": f"configs[2]: {N} chunks, dense IP top-{depth} + BM25 TAAT top-{depth} + RRF -> top-{TOPK}, {nq} queries per call",
                         "value": round(nq / t_all, 1), "unit": "queries/s", "legs_overlapped_equal_sequential": same_fused,
                         "dense_top50_qps": round(nq / t_dense, 1),
                         "bm25_qps": round(nq / t_sparse, 1), "rrf_qps": round(nq / t_fuse, 1),
                         "postings": int(postings.offsets[-1]), "postings_per_query": int(post_per_q),
                         "postings_build_s": round(build_s, 1),
                         "dense_top50_stats": dense_stats,
                         "bm25_roofline": {"bound": "lds/issue", "kernel": "taat_tile_kernel",
                                           "note": "not an HBM-bound kernel: a workgroup's time goes into LDS read-modify-write round "
                                                   "trips and slot barriers (DESIGN 5); the figures below are the posting bytes the "
                                                   "queries REQUEST (sum of df x 8 B) per second, most of which L2 serves",
                                           "requested_GBs": round(posting_gbs, 1), "requested_over_hbm_peak": round(posting_gbs / HBM_PEAK_GBS, 4),
                                           "traffic_from_profile": fetch}}
        del bm25, postings
        torch.cuda.empty_cache()

    def leg_encoder():
        from hiprag import EncoderConfig, HipEncoder
        cfg = EncoderConfig()                                        # XLM-R large: 24 x [H 1024, 16 heads, F 4096], vocab 250002
        enc = HipEncoder(cfg, seed=0, device=dev.index)
        rng = np.random.default_rng(0)
        bs, seq = 256, 512
        toks = [[0] + rng.integers(3, cfg.vocab, size=seq - 2).tolist() + [2] for _ in range(bs)]
        dt = _time_steps(torch, lambda: enc.encode_tokens(toks, batch_size=bs), 3, 1)
        flops = enc.last_flops()
        out["encoder"] = {"workload": "configs[4] embed stage: embed_batch(256 x 512 tokens), XLM-R-large shape, seeded random weights",
                          "value": round(bs * seq / dt, 1), "unit": "tokens/s", "ms_per_batch": round(dt * 1e3, 2),
                          "flops_per_batch": flops,
                          "roofline": {"bound": "mfma", "achieved": round(flops / dt / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                                       "frac": round(flops / dt / 2.5e15, 4),
                                       "note": "whole forward by wall time (host tokens in, embeddings on the GPU out)"}}
        del enc
        torch.cuda.empty_cache()

    # a leg that fails must not cost the headline line: its entry carries the error instead
    for name, fn in (("fp32", leg_fp32), ("reference", leg_reference), ("ivf", leg_ivf), ("hybrid", leg_hybrid), ("encoder", leg_encoder)):
        if name not in legs:
            continue
        try:
            fn()
        except Exception as e:
            import traceback
            print(f"[bench] leg {name} failed: {e!r}\n{traceback.format_exc()}", file=sys.stderr)
            out[{"fp32": "fp32_stream", "reference": "reference_shape"}.get(name, name)] = {"error": repr(e)}
            torch.cuda.empty_cache()
    return out


def run_config3(torch, dist, args, world, rank, local_rank, dev, use_dist):
    """BASELINE configs[3]: 10M x 1024-d rows AND the BM25 postings of the same 10M documents, both sharded by the same
    contiguous document ranges over the ranks; a step = one batch of hybrid queries through ShardedHybrid: local dense
    top-50 + local BM25 top-50 -> ONE all-gather of both packed partial lists -> global merges -> RRF -> top-10.
    Every rank builds only its own shard; the collection-wide BM25 constants (document frequencies, document count, average
    length) are agreed with one all-reduce at build time, so a shard's impacts equal the unsharded ones."""
    from collections import deque
    from hiprag import HipBM25, HipFlatIndex, build_postings
    from hiprag.sharded import ShardedHybrid, chunks_of_rank
    n_rows = args.rows if args.rows is not None else 10_000_000
    V, depth, terms_per_query = 262144, 50, 6
    n_chunks = 32
    chunk = (n_rows + n_chunks - 1) // n_chunks
    my_chunks = [c for c in chunks_of_rank(n_chunks, world, rank) if c * chunk < n_rows]
    row_lo = my_chunks[0] * chunk if my_chunks else 0
    n_local = sum(min(chunk, n_rows - c * chunk) for c in my_chunks)
    t0 = time.time()
    index = HipFlatIndex(DIM, "ip", device=local_rank)
    index.reserve_rows(n_local)
    cdf = torch.cumsum(1.0 / torch.arange(1, V + 1, dtype=torch.float64, device=dev), 0)
    cdf /= cdf[-1].clone()
    docs_h, terms_h, len_h = [], [], []
    for c in my_chunks:
        rows = min(chunk, n_rows - c * chunk)
        index.add_device(gen_chunk(torch, c, rows, dev))
        i = torch.arange(c * chunk, c * chunk + rows, dtype=torch.int64, device=dev)
        dl = 64 + (i * 2654435761) % 256
        g = torch.Generator(device=dev)
        g.manual_seed(777 + c)
        u = torch.rand(int(dl.sum().item()), generator=g, device=dev, dtype=torch.float64)
        terms_h.append(torch.clamp(torch.searchsorted(cdf, u), max=V - 1).to(torch.int32).cpu().numpy())
        docs_h.append(torch.repeat_interleave(i - row_lo, dl).to(torch.int32).cpu().numpy())
        len_h.append(dl.cpu().numpy())
        del u, i, dl
        if rank == 0:
            print(f"[bench] config3: rank 0 generated chunk {c} ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
    doc_len = np.concatenate(len_h) if len_h else np.zeros(0, np.int64)
    term = np.concatenate(terms_h) if terms_h else np.zeros(0, np.int32)
    doc = np.concatenate(docs_h) if docs_h else np.zeros(0, np.int32)
    del terms_h, docs_h, len_h
    # collection-wide constants: df counts DOCUMENTS per term, so count unique (term, doc) pairs locally, then sum over ranks
    pair = np.unique(term.astype(np.int64) * np.int64(max(n_local, 1)) + doc.astype(np.int64))
    df = np.bincount(pair // max(n_local, 1), minlength=V).astype(np.int64)
    del pair
    tot = torch.tensor([int(doc_len.sum())], dtype=torch.int64, device=dev)
    df_t = torch.from_numpy(df).to(dev)
    if use_dist and world > 1:
        dist.all_reduce(df_t)
        dist.all_reduce(tot)
    postings = build_postings(doc, term, n_local, V, doc_len, df_global=df_t.cpu().numpy(), n_docs_global=n_rows,
                              avgdl_global=int(tot.item()) / n_rows)
    del doc, term
    bm25 = HipBM25(postings, device=local_rank)
    n_postings = int(postings.offsets[-1])
    hy = ShardedHybrid(index, bm25, row_lo, gather=gather_for(args.backend))
    torch.cuda.synchronize()
    build_s = time.time() - t0

    gq = torch.Generator(device=dev)
    gq.manual_seed(4321)
    queries = torch.randn((N_QUERIES, DIM), generator=gq, device=dev, dtype=torch.float32)
    queries /= queries.norm(dim=1, keepdim=True)
    rng = np.random.default_rng(888)
    w = 1.0 / np.arange(17, V + 1, dtype=np.float64)
    cdfq = np.cumsum(w) / w.sum()
    sq = []
    for _ in range(N_QUERIES):
        t = []
        while len(t) < terms_per_query:
            cnd = int(min(np.searchsorted(cdfq, rng.random()), len(cdfq) - 1)) + 16
            if cnd not in t:
                t.append(cnd)
        sq.append(np.asarray(t, dtype=np.uint32))
    BATCH = hy.max_pass
    nb = N_QUERIES // BATCH
    IN_FLIGHT = 4

    def run_steps(n, first):
        pending, last = deque(), None
        for s_ in range(n):
            b = (first + s_) % nb
            pending.append(hy.search_begin(queries[b * BATCH:(b + 1) * BATCH], sq[b * BATCH:(b + 1) * BATCH], depth, TOPK))
            if len(pending) >= IN_FLIGHT:
                last = hy.search_end(pending.popleft())
        while pending:
            last = hy.search_end(pending.popleft())
        return last

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup, 0)
    barrier()
    index.enable_timing(1)
    barrier()
    t1 = time.perf_counter()
    last = run_steps(args.steps, args.warmup)
    barrier()
    elapsed = time.perf_counter() - t1
    st = index.stats()
    index.enable_timing(False)
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    mine = (rank, int(index.ntotal), int(row_lo), int(BATCH), float(st["avg_scan_ms"]), int(st["bytes_per_pass"]),
            int(st["fallback_queries"]), n_postings, [int(v) for v in last[1][0].tolist()])
    seen = [mine]
    if use_dist:
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        seen = sorted(seen)
    bad = (sum(v[1] for v in seen) != n_rows or any(v[3] != BATCH for v in seen) or [v[0] for v in seen] != list(range(world))
           or any(v[8] != seen[0][8] for v in seen))          # every rank must hold the same fused list
    if rank != 0:
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(3 if bad else 0)
    if bad:
        print(f"[bench] FATAL: the ranks disagree or do not cover the collection: {seen}", file=sys.stderr)
        sys.exit(3)
    passes = (BATCH + index.pass_queries - 1) // index.pass_queries
    slow = max(seen, key=lambda v: v[4])
    bpl = slow[5] * passes
    ach = bpl / (slow[4] * 1e-3) / 1e9 if slow[4] > 0 else 0.0
    out = {"metric": "hybrid queries/sec, dense top-50 + BM25 top-50 + RRF -> top-10, 10M x 1024-d index + postings row-sharded",
           "value": round(args.steps * BATCH / elapsed, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None,
           "dtype": "bf16 x bf16 -> f32 filter scan + f64 re-score (dense), f32 impact sums (BM25), f32 RRF", "data": "synthetic",
           "config": {"workload": (f"configs[3]: {n_rows} x 1024-d rows + BM25 postings of the same documents, row-sharded over "
                                   f"{world} rank(s), hybrid top-{TOPK} at fusion depth {depth}"),
                      "rows": n_rows, "dim": DIM, "k": TOPK, "depth": depth, "queries_per_step": BATCH, "passes_per_step": passes,
                      "sharding": f"rows/{world}" if world > 1 else "none", "world": world,
                      "exchange": (f"1 all-gather of [2,2,{BATCH},{depth}] int64 per step" if use_dist else "none"),
                      "allgather_payload_bytes_per_rank": (4 * BATCH * depth * 8 if use_dist else 0),
                      "backend": (args.backend if use_dist else "none"), "steps_in_flight": IN_FLIGHT,
                      "rows_local": [v[1] for v in seen], "postings_local": [v[7] for v in seen]},
           "ranks_seen": len(seen), "fallback_queries": sum(v[6] for v in seen), "build_s": round(build_s, 1),
           "fused_lists_equal_on_all_ranks": True,
           "roofline": {"bound": "hbm", "kernel": "scan_bf16_kernel (dense leg of the slowest rank)", "achieved": round(ach, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                        "bytes_per_launch": bpl, "passes_per_launch": passes, "avg_launch_ms": round(slow[4], 5)}}
    print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def gather_for(backend):
    """None = the product's RCCL all_gather_into_tensor; the gloo rehearsals (two ranks sharing one GPU) need the list form."""
    if backend == "nccl":
        return None

    def list_gather(pack, gathered, group=None, async_op=True):
        import torch.distributed as dist
        return dist.all_gather([gathered[r] for r in range(gathered.shape[0])], pack, group=group, async_op=async_op)
    return list_gather


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, exactly as the driver
    would (python -m torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1), from a parent that never
    initialises the GPU.  Returns the children's exit code; rank 0's JSON line reaches stdout through the inherited pipe."""
    import socket
    import subprocess
    if not args.share_gpu:
        import torch                      # device_count() does not create a HIP context on this image
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"[bench] FATAL: --gpus {args.gpus} but only {have} GPU(s) are visible; one rank per GPU over RCCL needs "
                  f"{args.gpus} (one-GPU rehearsal: --share-gpu --backend gloo)", file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] no launcher around --gpus {args.gpus}: starting the ranks with torch.distributed.run", file=sys.stderr)
    return subprocess.run(cmd, env=env, cwd=REPO).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=None, help="override the index size (debug only; default 1M, config3: 10M)")
    ap.add_argument("--workload", default="config1", choices=["config1", "config3"],
                    help="config1 = BASELINE configs[1], the headline (dense top-10 on 1M rows, rows sharded over the ranks); "
                         "config3 = BASELINE configs[3]: 10M rows + BM25 postings row-sharded over the ranks, hybrid top-10 through "
                         "ShardedHybrid (one packed all-gather per step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--legs", default="fp32,reference,ivf,hybrid,encoder", help="extra single-GPU legs (N = 1 only): comma list or 'none'")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo; RCCL refuses duplicate GPUs)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal only: run the multi-GPU code path (process group, all-gather per step, merge) on ONE rank, "
                         "so that a one-GPU box executes the RCCL calls of the N > 1 run")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    # ---- N > 1 without a launcher: this process becomes the launcher.  It starts one fresh child per GPU through
    # torch.distributed.run BEFORE anything here has touched the GPU (no HIP call, no torch.cuda call other than the device
    # count, which does not create a context), relays their output (rank 0 prints the JSON line) and exits with their code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import torch
    import torch.distributed as dist
    from hiprag import HipFlatIndex
    from hiprag.sharded import ShardedFlatIndex, chunks_of_rank

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # a run that silently measures another number of GPUs than it was asked for is worse than no run
        if rank == 0:
            print(f"[bench] FATAL: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    use_dist = world > 1 or args.force_dist
    if args.force_dist and world == 1:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ["HIPRAG_FORCE_EXCHANGE"] = "1"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the one collective of a step moves 80-160 KB per rank: four channels are plenty, and every channel is a workgroup
        # that spins on a CU until all ranks have joined -- CUs the scan (which leaves 8 spare) would otherwise wait for
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "4")
        if args.share_gpu:
            local_rank = 0
        elif local_rank >= torch.cuda.device_count():
            print(f"[bench] FATAL: rank {rank} wants cuda:{local_rank} but only {torch.cuda.device_count()} GPU(s) are visible "
                  f"(RCCL needs one GPU per rank; --share-gpu --backend gloo is the one-GPU rehearsal)", file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
        if dist.get_world_size() != args.gpus:
            print(f"[bench] FATAL: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}", file=sys.stderr)
            sys.exit(2)
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    if args.workload == "config3":
        return run_config3(torch, dist, args, world, rank, local_rank, dev, use_dist)
    if args.rows is None:
        args.rows = N_ROWS
    n_rows = args.rows
    chunk = CHUNK if n_rows == N_ROWS else max(1, n_rows // 32)

    # ---- build the local shard (rows resident in HBM before anything is timed) ----------------------
    n_chunks = (n_rows + chunk - 1) // chunk
    my_chunks = chunks_of_rank(n_chunks, world, rank)
    row_lo = my_chunks[0] * chunk if my_chunks else 0
    index = HipFlatIndex(DIM, "ip", device=local_rank)
    if os.environ.get("HIPRAG_BENCH_NO_RESERVE") != "1":
        index.reserve_rows(sum(min(chunk, n_rows - c * chunk) for c in my_chunks))   # the shard's size is known: no growth cycles
    keep_host = (world == 1 and not args.no_cpu_baseline)
    host_rows = []
    t0 = time.time()
    for c in my_chunks:
        rows = min(chunk, n_rows - c * chunk)
        x = gen_chunk(torch, c, rows, dev)
        index.add_device(x)
        if keep_host:
            host_rows.append(x.cpu().numpy())
        del x
    torch.cuda.synchronize()
    build_s = time.time() - t0
    sharded = ShardedFlatIndex(index, row_lo, gather=gather_for(args.backend))
    index.reserve_search(TOPK)

    gq = torch.Generator(device=dev)
    gq.manual_seed(4321)
    queries = torch.randn((N_QUERIES, DIM), generator=gq, device=dev, dtype=torch.float32)
    queries /= queries.norm(dim=1, keepdim=True)
    # one scan LAUNCH per step: 512 queries = 8 passes of 64 run back to back inside the launch (each pass streams the
    # bf16 copy of the local rows once for its own query tile).  The library sizes a launch by the LOCAL row count so that
    # it lasts about as long whatever the shard: 8 passes at 1M rows, 16 (1024 queries) at <= 500k rows per GPU.
    # HIPRAG_LAUNCH_QUERIES / HIPRAG_SCAN_MODE change the split
    BATCH = sharded.max_pass          # = index.launch_queries on one GPU; the MINIMUM over the ranks otherwise (agreed at construction)
    PASSES = (BATCH + index.pass_queries - 1) // index.pass_queries
    nb = N_QUERIES // BATCH

    IN_FLIGHT = 4      # steps in flight (the library has 8 workspace slots); results are consumed in order

    def run_steps(n, first):
        """n pipelined steps: while step i scans, the tails (select / re-score / all-gather / merge) of the previous
        steps run on their own streams."""
        from collections import deque
        pending, last = deque(), None
        for s in range(n):
            b = (first + s) % nb
            pending.append(sharded.search_begin(queries[b * BATCH:(b + 1) * BATCH], TOPK))
            if len(pending) >= IN_FLIGHT:
                last = sharded.search_end(pending.popleft())
        while pending:
            last = sharded.search_end(pending.popleft())
        return last

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup, 0)
    barrier()
    # HIP events around EVERY scan launch of the timed region.  (Sampling every n-th launch is possible -- enable_timing(n) --
    # and saves ~20 us of dispatch bubble per launch, but it biases the figure: a bracketed launch is dispatched a little
    # later than its unbracketed neighbours, loses the race for CUs against the tail kernels of earlier steps, and
    # measures ~3 % longer than the in-kernel stamps of the same launches.)
    index.enable_timing(1)
    barrier()
    t0 = time.perf_counter()
    last = run_steps(args.steps, args.warmup)
    barrier()
    elapsed = time.perf_counter() - t0
    st = index.stats()
    index.enable_timing(False)
    scan_ms, wall_ms = float(st["avg_scan_ms"]), float(st["avg_scan_wall_ms"])
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- p50 latency of single queries: (a) through the host boundary -- hipidx_search with HOST arrays in and out: H2D of
    # the query, scan, tails, D2H of the result, sync -- which is what BASELINE.md / DESIGN.md quote; (b) device-resident -----
    lat, lat_dev = [], []
    q_host = queries[:120].cpu().numpy()
    for i in range(120):
        t1 = time.perf_counter()
        index.search(q_host[i:i + 1], TOPK)
        lat.append((time.perf_counter() - t1) * 1e3)
        qi = queries[i:i + 1]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sharded.search_device(qi, TOPK)
        torch.cuda.synchronize()
        lat_dev.append((time.perf_counter() - t1) * 1e3)
    lat = np.sort(np.asarray(lat[20:]))
    lat_dev = np.sort(np.asarray(lat_dev[20:]))
    # the same index through ONE library call per batch of all N_QUERIES queries (hipidx_search_dev pipelines the launches of
    # such a batch itself: no Python between the steps, no timing events) -- the local shard only, no exchange
    one_call_qps = None
    if world == 1:
        for _ in range(2):
            index.search_device(queries, TOPK)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(6):
            index.search_device(queries, TOPK)
        torch.cuda.synchronize()
        one_call_qps = round(6 * N_QUERIES / (time.perf_counter() - t1), 1)

    rows_local = [int(index.ntotal)]
    per_rank = [{"rank": 0, "rows": int(index.ntotal), "scan_ms": round(scan_ms, 5), "scan_ms_gpu_clock": round(wall_ms, 5),
                 "bytes_per_pass": int(st["bytes_per_pass"]), "fallback_queries": int(st["fallback_queries"])}]
    if use_dist:      # what every rank holds and measured, as seen by the collective: lets a reader check that N ranks really took part
        seen = [None] * world
        dist.all_gather_object(seen, (rank, int(index.ntotal), int(row_lo), int(sharded.max_pass), scan_ms, wall_ms,
                                      int(st["bytes_per_pass"]), int(st["fallback_queries"])))
        seen = sorted(seen)
        rows_local = [v[1] for v in seen]
        if sum(rows_local) != n_rows or any(v[3] != sharded.max_pass for v in seen) or [v[0] for v in seen] != list(range(world)):
            if rank == 0:
                print(f"[bench] FATAL: the ranks do not cover the index: {seen}", file=sys.stderr)
            sys.exit(3)
        per_rank = [{"rank": v[0], "rows": v[1], "scan_ms": round(v[4], 5), "scan_ms_gpu_clock": round(v[5], 5),
                     "bytes_per_pass": v[6], "fallback_queries": v[7]} for v in seen]
    if rank != 0:
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    # N > 1: the roofline line is the SLOWEST rank's scan against the bytes that rank streams (the step ends with it)
    slow = max(per_rank, key=lambda r: r["scan_ms"])
    scan_ms, wall_ms = slow["scan_ms"], slow["scan_ms_gpu_clock"]

    qps = args.steps * BATCH / elapsed
    # local rows * d_pad * 2 per pass (the bf16 filter copy), PASSES passes per launch: what ONE scan launch streams
    bytes_per_launch = int(slow["bytes_per_pass"]) * PASSES
    achieved = bytes_per_launch / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    out = {
        "metric": "queries/sec, exact top-10 inner-product search, 1M x 1024-d fp32 index",
        "value": round(qps, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "bf16 x bf16 -> f32 filter scan, f64 re-score of the f32 rows",
        "data": "synthetic",
        "config": {"workload": ("configs[1]: 1M x 1024-d synthetic unit vectors, brute-force inner-product top-10" if n_rows == N_ROWS
                                else f"--rows override: {n_rows} x 1024-d synthetic unit vectors, brute-force inner-product top-10"),
                   "rows": n_rows, "dim": DIM, "k": TOPK, "queries_per_step": BATCH, "passes_per_step": PASSES,
                   "queries_per_pass": index.pass_queries,
                   "sharding": f"rows/{world}" if world > 1 else "none",
                   "exchange": f"1 all-gather of [2,{BATCH},{TOPK}] int64 per step" if use_dist else "none",
                   "steps_in_flight": IN_FLIGHT, "world": world, "rows_local": rows_local,
                   "backend": (args.backend if use_dist else "none"),
                   "allgather_payload_bytes_per_rank": (2 * BATCH * TOPK * 8 if use_dist else 0)},
        "p50_ms_single_query": round(float(lat[len(lat) // 2]), 4),
        "p99_ms_single_query": round(float(lat[int(len(lat) * 0.99) - 1]), 4),
        "p50_ms_single_query_device_resident": round(float(lat_dev[len(lat_dev) // 2]), 4),
        "latency_path": "hipidx_search: host query in, host results out (H2D + scan + tails + D2H + sync) on the local shard",
        "one_hipidx_search_dev_call_per_4096_queries_qps": one_call_qps,
        "fallback_queries": sum(r["fallback_queries"] for r in per_rank),
        "finish_work_rank0": {"extended_queries_share": round(st["roundb_queries"] / max(1, st["queries"]), 4),
                              "list_entries_per_query": round(st["list_entries"] / max(1, st["queries"]), 1),
                              "ranked_entries_per_query": round(st["ranked_entries"] / max(1, st["queries"]), 1),
                              "rescored_groups_per_query": round(st["rescored_groups"] / max(1, st["queries"]), 1)},
        "ranks_seen": len(per_rank),
        "per_rank": per_rank,
        "scan_ms_min_max": [min(r["scan_ms"] for r in per_rank), max(r["scan_ms"] for r in per_rank)],
        "build_s": round(build_s, 2),
        "roofline": {"bound": "hbm", "kernel": "scan_bf16_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "bytes_per_launch": bytes_per_launch, "passes_per_launch": PASSES,
                     "avg_launch_ms": round(scan_ms, 5), "launches_timed": int(st["timed_passes"]),
                     "avg_launch_ms_gpu_clock": round(wall_ms, 5), "avg_gap_ms_between_launches": round(float(st["avg_scan_gap_ms"]), 5)},
    }
    # calibration (outside the timed region): what THIS device, now, streams read-only with the scan's partition -- boxes
    # of the pool and allocations inside one process differ by several percent
    try:
        import ctypes
        from hiprag import _native as nat
        gbps = ctypes.c_double()
        nat.call("hiprag_probe_read_gbps", local_rank, 2_048_000_000, 5, ctypes.byref(gbps))
        out["roofline"]["read_probe_GBs"] = round(gbps.value, 1)
        out["roofline"]["frac_of_read_probe"] = round(achieved / gbps.value, 4)
    except Exception as e:   # a diagnostic must never cost the line
        out["roofline"]["read_probe_GBs"] = None
        print(f"[bench] read probe failed: {e!r}", file=sys.stderr)

    # ---- CPU baseline: the oracle's reference-faithful twin, bounded sample, rank 0, N=1 only ----------
    if world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import hybrid_oracle as ho
            xh = np.concatenate(host_rows, axis=0)
            del host_rows
            nwarm, nsample = 20, 200                                              # 20 warm-up + 200 timed queries
            qh = queries[nwarm:nwarm + nsample].cpu().numpy()
            ho.flat_search_f32_faithful(xh, queries[:nwarm].cpu().numpy(), TOPK, ho.METRIC_IP)
            lat_cpu = []
            ci = np.empty((nsample, TOPK), dtype=np.int64)
            t1 = time.perf_counter()
            for i in range(nsample):                                              # one query per call, like the reference (faiss_index.py:81-83)
                t2 = time.perf_counter()
                ci[i] = ho.flat_search_f32_faithful(xh, qh[i:i + 1], TOPK, ho.METRIC_IP)[1][0]
                lat_cpu.append(time.perf_counter() - t2)
            cpu_s = time.perf_counter() - t1
            lat_cpu = np.sort(np.asarray(lat_cpu)) * 1e3
            g64, g32, gi = index.search_device(queries[nwarm:nwarm + nsample], TOPK)
            torch.cuda.synchronize()
            agree = bool(np.array_equal(gi.cpu().numpy(), ci))
            out["cpu_baseline"] = {"value": round(nsample / cpu_s, 3), "unit": "queries/s", "cores": 1, "kind": "port",
                                   "sample": f"{nwarm} warm-up + {nsample} timed queries x full {n_rows}x{DIM} index, one query per call, "
                                             f"1 thread, fp32 C restatement of IndexFlat search (FAISS itself is not installed)",
                                   "p50_ms": round(float(lat_cpu[nsample // 2]), 2), "p99_ms": round(float(lat_cpu[int(nsample * 0.99) - 1]), 2),
                                   "host_cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)),
                                   "ids_equal_gpu": agree}
            # SURVEY 8d row (ii), reported beside the faithful port: best-effort CPU -- one batched X @ Q^T on numpy's BLAS
            # (all host cores) + argpartition; fp32 BLAS summation order, so a near-tie may rank differently from the fp64 truth
            nb = 64
            qb = queries[:nb].cpu().numpy()
            (xh[:50000] @ qb.T).shape                                           # warm BLAS threads
            t2 = time.perf_counter()
            sc = xh @ qb.T                                                       # [N, nb]
            part = np.argpartition(-sc, TOPK, axis=0)[:TOPK]                     # [TOPK, nb], unordered
            top = np.take_along_axis(sc, part, axis=0)
            order = np.lexsort((part, -top), axis=0)
            be_ids = np.take_along_axis(part, order, axis=0).T                   # [nb, TOPK]
            be_s = time.perf_counter() - t2
            gbi = index.search_device(queries[:nb], TOPK)[2].cpu().numpy()
            out["cpu_baseline"]["best_effort"] = {"value": round(nb / be_s, 2), "unit": "queries/s",
                                                  "cores": len(os.sched_getaffinity(0)),
                                                  "sample": f"{nb} queries in one batch: numpy BLAS X @ Q^T on every host core + "
                                                            f"argpartition top-{TOPK}",
                                                  "ids_equal_gpu_fraction": round(float(np.mean(be_ids == gbi)), 4)}
        except Exception as e:   # the baseline must not cost the GPU line
            import traceback
            print(f"[bench] cpu_baseline failed: {e!r}\n{traceback.format_exc()}", file=sys.stderr)
            out.setdefault("cpu_baseline", {})["error"] = repr(e)
    # HBM traffic of the scan kernel comes from a separate rocprofv3 --pmc pass (counters cannot be read from inside this
    # process); the committed summary applies to the full-size single-GPU workload only.
    pmc = os.path.join(REPO, "profiles", "r03_pmc_scan.json")
    if world == 1 and n_rows == N_ROWS and os.path.exists(pmc):
        with open(pmc) as f:
            p = json.load(f)
        # NOT measured in this run: counters cannot be read from inside the process.  `traffic` stays null; the number of the
        # committed PMC summary (same binary path, same workload) travels under a key that says where it comes from.
        out["roofline"]["traffic_from_profile"] = {"bytes_per_launch": int(p["traffic_bytes_per_launch"]),
                                                   "source": "profiles/r03_pmc_scan.json (rocprofv3 --pmc FETCH_SIZE x2 + "
                                                             "WRITE_SIZE, separate passes; not collected by this run)"}
    if world == 1 and args.legs != "none":
        del sharded
        out["legs"] = run_legs(torch, args, dev, index, queries, [v.strip() for v in args.legs.split(",") if v.strip()])
    print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
