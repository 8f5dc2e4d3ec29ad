#!/usr/bin/env python3
"""Debug probe: scan kernel time (HIP events) for single-pass and 8-pass launches."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "intool-rag_amd"))
import torch
from hiprag import HipFlatIndex
dev = torch.device("cuda", 0)
ix = HipFlatIndex(1024, "ip")
for c in range(32):
    g = torch.Generator(device=dev); g.manual_seed(1234 + c)
    x = torch.randn((31250, 1024), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
    ix.add_device(x)
g = torch.Generator(device=dev); g.manual_seed(4321)
q = torch.randn((4096, 1024), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
for nq in (64, 512):
    for _ in range(3): ix.search_device(q[:nq], 10)
    torch.cuda.synchronize()
    ix.enable_timing(1)
    for i in range(10): ix.search_device(q[i * 64:i * 64 + nq], 10)
    torch.cuda.synchronize()
    st = ix.stats(); ix.enable_timing(False)
    print(f"nq {nq:4d}: scan {st['avg_scan_ms']:.3f} ms (gpu clock {st['avg_scan_wall_ms']:.3f})", flush=True)
