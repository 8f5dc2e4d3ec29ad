#!/usr/bin/env python3
"""Debug probe of the scan's in-kernel candidate filter: list lengths and launch times, launch by launch."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "intool-rag_amd"))
import torch
from hiprag import HipFlatIndex
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
ix = HipFlatIndex(1024, "ip")
chunk = 31250
for c in range(rows // chunk):
    g = torch.Generator(device=dev); g.manual_seed(1234 + c)
    x = torch.randn((chunk, 1024), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
    ix.add_device(x)
g = torch.Generator(device=dev); g.manual_seed(4321)
q = torch.randn((4096, 1024), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
B = ix.launch_queries
prev = ix.stats()
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix.search_device(q[(it % 8) * B:(it % 8 + 1) * B], k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    st = ix.stats()
    nq = st["queries"] - prev["queries"]
    print(f"launch {it}: {dt:8.3f} ms  list/q {(st['list_entries']-prev['list_entries'])/nq:9.1f}  ranked/q {(st['ranked_entries']-prev['ranked_entries'])/nq:7.1f}  "
          f"rescored/q {(st['rescored_groups']-prev['rescored_groups'])/nq:6.1f}  extended {st['roundb_queries']-prev['roundb_queries']:4d}  fallback {st['fallback_queries']-prev['fallback_queries']:4d}", flush=True)
    prev = st
