#!/usr/bin/env python3
"""A few hybrid calls (config 2 shapes) for a rocprofv3 --kernel-trace timeline: which kernels of the two legs overlap."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "intool-rag_amd"))
import torch
from hiprag import HipBM25, HipFlatIndex, build_postings, hybrid_search_device

dev = torch.device("cuda", 0)
N, V, d, nq, depth, k = 1_000_000, 262144, 1024, 256, 50, 10
index = HipFlatIndex(d, "ip")
for c in range(0, N, 125000):
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + c // 125000)
    x = torch.randn((125000, d), generator=g, device=dev)
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
torch.cuda.synchronize()
for _ in range(int(os.environ.get("CALLS", 6))):
    hybrid_search_device(index, bm25, queries, sq, depth=depth, k=k)
torch.cuda.synchronize()
print("done", flush=True)
