#!/usr/bin/env python3
"""Debug probe: scan state (count, thetac, slots) after a scan launch without its finish."""
import ctypes, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "intool-rag_amd"))
import torch
from hiprag import HipFlatIndex, _native as nat
rows = 1_000_000
dev = torch.device("cuda", 0)
ix = HipFlatIndex(1024, "ip")
chunk = 31250
for c in range(rows // chunk):
    g = torch.Generator(device=dev); g.manual_seed(1234 + c)
    x = torch.randn((chunk, 1024), generator=g, device=dev); x /= x.norm(dim=1, keepdim=True)
    ix.add_device(x)
g = torch.Generator(device=dev); g.manual_seed(4321)
q = torch.randn((4096, 1024), generator=g, device=dev); q /= q.norm(dim=1, keepdim=True)
lib = nat.load()
ix.enable_timing(1)
def unord(o):
    o = o.astype(np.uint32)
    b = np.where(o & 0x80000000, o ^ 0x80000000, o ^ 0xFFFFFFFF).astype(np.uint32)
    return b.view(np.float32)
for nq in (64, 128, 512):
    ix.search_begin(q[:nq], 10, slot=1)
    torch.cuda.synchronize()
    Q = 1024
    cnt = np.zeros(Q, np.uint32); th = np.zeros(Q, np.uint32); sl = np.zeros(16 * 4096, np.uint32); oq = ctypes.c_int32(); arr = np.zeros(16, np.uint32)
    rc = lib.hipidx_debug_scan_state(ctypes.c_uint64(ix._h), 1, cnt.ctypes.data_as(ctypes.c_void_p), th.ctypes.data_as(ctypes.c_void_p), sl.ctypes.data_as(ctypes.c_void_p), ctypes.byref(oq), arr.ctypes.data_as(ctypes.c_void_p))
    Qr = oq.value
    sl = sl[:(Qr // 64) * 4096].reshape(Qr // 64, 64, 64)
    st = ix.stats()
    print(f"== nq {nq} (workspace Q {Qr}) wait timeouts {arr[1]} recompute m==0 {arr[2]} ok {arr[3]} staged {arr[4]} stage-overflow {arr[5]} waits {arr[6]} wait-ticks {arr[7]}  scan_ms {st['avg_scan_ms']:.3f} wall {st['avg_scan_wall_ms']:.3f}")
    for p in range((nq + 63) // 64):
        c = cnt[p * 64:(p + 1) * 64]; t = th[p * 64:(p + 1) * 64]
        smin = sl[p].min(axis=0)
        zero_slots = int((sl[p] == 0).sum())
        print(f" pass {p}: count min/med/max {c.min()} {int(np.median(c))} {c.max()}  thetac==slotmin for {(t == smin).sum()}/64 queries, thetac zero for {(t == 0).sum()}, zero slots {zero_slots}  theta[0] {unord(t[:1])[0]:.4f} slotmin[0] {unord(smin[:1])[0]:.4f}")
    # finish to clean the state
    out = (torch.empty((nq, 10), dtype=torch.float64, device=dev), torch.empty((nq, 10), dtype=torch.float32, device=dev), torch.empty((nq, 10), dtype=torch.int64, device=dev))
    ix.search_finish(q[:nq], 10, 1, out)
    torch.cuda.synchronize()
