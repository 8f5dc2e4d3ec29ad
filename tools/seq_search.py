"""Sequential (non-pipelined) hipidx_search_dev loop on a synthetic 1M x 1024 index, one full launch (256 queries = 4
passes by default) at a time: the command the rocprofv3 --pmc and --kernel-trace profiles under profiles/ are taken
with (ROWS=<n> overrides the size, HIPRAG_LAUNCH_QUERIES / HIPRAG_SCAN_MODE the launch shape)."""
import sys, os, time, numpy as np, torch
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'intool-rag_amd'))
from hiprag import HipFlatIndex
N=int(os.environ.get('ROWS','1000000')); d=1024; k=10
g=torch.Generator(device='cuda'); g.manual_seed(1234)
ix=HipFlatIndex(d,'ip')
for lo in range(0,N,125000):
    m=min(125000,N-lo)
    x=torch.randn((m,d),generator=g,device='cuda',dtype=torch.float32); x/=x.norm(dim=1,keepdim=True); ix.add_device(x)
B=ix.launch_queries   # AFTER the rows are in: exactly one launch per call (a larger batch would be pipelined by the library)
q=torch.randn((B,d),generator=g,device='cuda'); q/=q.norm(dim=1,keepdim=True)
out=None
for _ in range(5): out=ix.search_device(q,k,out)
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(30): ix.search_device(q,k,out)
torch.cuda.synchronize(); print('sequential ms/launch %.4f (%d queries, %d passes)'%((time.perf_counter()-t0)/30*1e3,B,B//ix.pass_queries))
