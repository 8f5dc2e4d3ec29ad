#!/usr/bin/env python3
"""
Is the dense search rate a property of the box, of time, or of where a process's allocations landed?  One process builds
the 1M x 1024 index several times (fresh allocations each time; the previous index is freed first) and measures, per build:
the pipelined rate (bench.py's loop, 2 x 300 steps) and one launch at a time (scan duration by the library's HIP events).
Round 2's answer: constant to 0.1 % inside a build, up to 14 % apart between builds of the same process.
usage: python tools/alloc_modes.py [builds]
"""
import os
import sys
import time
from collections import deque

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "intool-rag_amd"))


def main():
    import torch
    import bench
    from hiprag import HipFlatIndex
    from hiprag.sharded import ShardedFlatIndex
    builds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda", 0)
    gq = torch.Generator(device=dev)
    gq.manual_seed(4321)
    queries = torch.randn((4096, 1024), generator=gq, device=dev)
    queries /= queries.norm(dim=1, keepdim=True)
    for build in range(builds):
        ix = HipFlatIndex(1024, "ip", device=0)
        for c in range(8):
            ix.add_device(bench.gen_chunk(torch, c, 125000, dev))
        torch.cuda.synchronize()
        sh = ShardedFlatIndex(ix, 0)
        ix.reserve_search(10)
        B = sh.max_pass
        nb = 4096 // B

        def run(n):
            pend = deque()
            for s in range(n):
                b = s % nb
                pend.append(sh.search_begin(queries[b * B:(b + 1) * B], 10))
                if len(pend) >= 4:
                    sh.search_end(pend.popleft())
            while pend:
                sh.search_end(pend.popleft())

        run(20)
        torch.cuda.synchronize()
        rates = []
        for _ in range(2):
            t0 = time.perf_counter()
            run(300)
            torch.cuda.synchronize()
            rates.append(round(300 * B / (time.perf_counter() - t0) / 1e3, 1))
        qq, res = queries[:B], None
        for _ in range(5):
            res = ix.search_device(qq, 10, res)
        torch.cuda.synchronize()
        ix.enable_timing(1)
        t0 = time.perf_counter()
        for _ in range(30):
            ix.search_device(qq, 10, res)
        torch.cuda.synchronize()
        seq_ms = (time.perf_counter() - t0) / 30 * 1e3
        scan_ms = ix.stats()["avg_scan_ms"]
        ix.enable_timing(False)
        print(f"build {build}: pipelined {rates} k queries/s | one launch at a time {seq_ms:.3f} ms, scan {scan_ms:.3f} ms", flush=True)
        del sh
        ix.close()
        del ix
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
