#!/usr/bin/env python3
"""
Headline shape (1M x 1024, top-10, launches of launch_queries queries chained like bench.py): rate and scan time for
every number of CUs the scan leaves free (hipidx_set_spare_cus), with the finish in stream order and beside the next scan
(on the CUs the scan left).  ROWS / K / SPARES from the environment.  One JSON line per setting.
"""
import json
import os
import sys
import time
from collections import deque

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "intool-rag_amd"))


def main():
    import torch
    from hiprag import HipFlatIndex
    from hiprag.sharded import ShardedFlatIndex
    dev = torch.device("cuda", 0)
    rows, k, d = int(os.environ.get("ROWS", 1_000_000)), int(os.environ.get("K", 10)), 1024
    pre = os.environ.get("PRE", "none")          # a large torch allocation BEFORE the index buffers: none | keep | free
    chunk = int(os.environ.get("CHUNK", 125000))  # rows per add
    dummy = None
    if pre != "none":
        dummy = torch.empty(int(os.environ.get("PRE_MB", 512)) << 20, dtype=torch.uint8, device=dev)
        if pre == "free":
            del dummy
            torch.cuda.empty_cache()
    ix = HipFlatIndex(d, os.environ.get("METRIC", "ip"))
    if os.environ.get("RESERVE") == "1":
        ix.reserve_rows(rows)
    for c in range(0, rows, chunk):
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + c // chunk)
        x = torch.randn((min(chunk, rows - c), d), generator=g, device=dev)
        x /= x.norm(dim=1, keepdim=True)
        ix.add_device(x)
    del x
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    queries = torch.randn((8192, d), generator=g, device=dev)
    queries /= queries.norm(dim=1, keepdim=True)
    ix.reserve_search(k)
    batch = ix.launch_queries
    nb = queries.shape[0] // batch
    ref = None
    skipped = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("SKIP_STREAMS", 0)))]   # dummy streams created first
    if os.environ.get("USE_SKIPPED") == "1":
        for st_ in skipped:
            with torch.cuda.stream(st_):
                torch.zeros(8, device=dev)
        torch.cuda.synchronize()
    tail_prio = os.environ.get("TAIL_PRIO")      # priority of the tail streams (HIP: -1 high, 0 normal, 1 low)
    tails = []
    if tail_prio is not None:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        lo, hi = ctypes.c_int(), ctypes.c_int()
        hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
        print(json.dumps({"priority_range_least_greatest": [lo.value, hi.value]}), flush=True)
        for _ in range(2):
            st = ctypes.c_void_p()
            rc = hip.hipStreamCreateWithPriority(ctypes.byref(st), 1, int(tail_prio))
            assert rc == 0, rc
            tails.append(torch.cuda.ExternalStream(st.value, device=dev))
    if os.environ.get("SCAN_PRIO") == "high":
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    for spare in [int(s) for s in os.environ.get("SPARES", "0,16,32,48,64,96").split(",")]:
        for aside in ((True,) if os.environ.get("ASIDE_ONLY") == "1" else (False, True)):
            sh = ShardedFlatIndex(ix, 0, tails_aside=aside)
            layout = os.environ.get("CU_MASK")       # interleaved | blocked: scan and tail streams on disjoint CU sets
            if aside and layout:
                import ctypes
                hip = ctypes.CDLL("libamdhip64.so")
                ncu = 256
                tail_bits = [i for i in range(ncu) if ((i // 8) >= (ncu - spare) // 8 if layout == "interleaved" else i >= ncu - spare)]
                def mk(bits):
                    words = (ctypes.c_uint32 * (ncu // 32))()
                    for b in bits:
                        words[b // 32] |= (1 << (b % 32))
                    st = ctypes.c_void_p()
                    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ncu // 32, words)
                    assert rc == 0, rc
                    return torch.cuda.ExternalStream(st.value, device=dev)
                scan_bits = [i for i in range(ncu) if i not in set(tail_bits)]
                sh.scan = mk(scan_bits)
                tails2 = [mk(tail_bits), mk(tail_bits)]
                sh.side = [tails2[i % 2] for i in range(len(sh.side))]
                sh._side_ptr = [st.cuda_stream for st in sh.side]
            if aside and tail_prio is not None:
                sh.side = [tails[i % 2] for i in range(len(sh.side))]
                sh._side_ptr = [st.cuda_stream for st in sh.side]
            ix.set_spare_cus(spare)

            host = {"begin": 0.0, "end": 0.0, "n": 0}

            def steps(n, first=0):
                pending = deque()
                last = None
                for s in range(n):
                    b = (first + s) % nb
                    h0 = time.perf_counter()
                    pending.append(sh.search_begin(queries[b * batch:(b + 1) * batch], k))
                    h1 = time.perf_counter()
                    host["begin"] += h1 - h0
                    host["n"] += 1
                    if len(pending) >= 4:
                        last = sh.search_end(pending.popleft())
                        host["end"] += time.perf_counter() - h1
                while pending:
                    last = sh.search_end(pending.popleft())
                return last
            steps(6)
            torch.cuda.synchronize()
            ix.enable_timing(1)
            torch.cuda.synchronize()
            n = int(os.environ.get("STEPS", 40))
            t0 = time.perf_counter()
            steps(n, 6)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            st = ix.stats()
            ix.enable_timing(False)
            got = steps(1, 0)
            torch.cuda.synchronize()
            ids = got[2].clone()
            if ref is None:
                ref = ids
            print(json.dumps({"rows": rows, "k": k, "batch": batch, "spare_cus": spare, "finish_beside_next_scan": aside,
                              "qps": round(n * batch / el, 1), "ms_per_step": round(el / n * 1e3, 4),
                              "scan_ms": round(st["avg_scan_ms"], 4), "scan_ms_gpu_clock": round(st["avg_scan_wall_ms"], 4),
                              "gap_ms": round(st["avg_scan_gap_ms"], 4), "ids_equal": bool(torch.equal(ids, ref)),
                              "host_ms_per_step_in_begin": round(host["begin"] / max(1, host["n"]) * 1e3, 4),
                              "host_ms_per_step_in_end": round(host["end"] / max(1, host["n"]) * 1e3, 4)}), flush=True)


if __name__ == "__main__":
    main()
