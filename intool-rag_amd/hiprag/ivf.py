"""
HipIVFIndex -- IVF-Flat on top of the flat index (BASELINE north_star names "the flat-IP / IVF distance scan"; the reference
itself only ever builds faiss.IndexFlatL2, rag/storage/faiss_index.py:123).  The approximate, low-latency mode for ONE query at
a time: a search reads nprobe / nlist of the rows instead of all of them.  For batches the flat index is the faster AND exact
choice (64 queries share one read of its 2-byte filter copy; an IVF probe is per query), see DESIGN.md.

Build (host side, here): k-means over the rows -- assignment = the flat index's own exact k = 1 search among the centroids
(libhiprag), centroid update = a segment mean (torch, plumbing) -- then the rows are stored permuted by list in an ordinary
HipFlatIndex, every list padded to whole 32-row blocks.  Search: libhiprag's hipivf_search_dev (csrc/dense_index.hip).
At nprobe = nlist every row is scored and the result equals the flat index's bit for bit.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np

from . import _native as nat
from .index import HipFlatIndex, _METRICS, _stream_ptr


class HipIVFIndex:
    def __init__(self, d: int, nlist: int, metric="l2", device: int = 0):
        self.d, self.nlist, self.metric, self.device = int(d), int(nlist), _METRICS[metric], int(device)
        self.rows: Optional[HipFlatIndex] = None
        self.centroids: Optional[HipFlatIndex] = None
        self._h = None
        self.ntotal = 0

    # ---- build ------------------------------------------------------------------------------------------------------
    def _assign(self, cents: HipFlatIndex, x, chunk: int = 65536):
        """nearest centroid of every row under the index's metric: the flat index's exact k = 1 search"""
        import torch
        out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
        for o in range(0, x.shape[0], chunk):
            out[o:o + chunk] = cents.search_device(x[o:o + chunk], 1)[2][:, 0]
        return out

    def train_add(self, x, iters: int = 6, seed: int = 0) -> None:
        """x: float32 [n, d] CUDA tensor or array -- trains the nlist centroids on x and stores x (ids = row numbers)."""
        import torch
        dev = torch.device("cuda", self.device)
        if not (hasattr(x, "is_cuda") and x.is_cuda):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
        n = x.shape[0]
        if x.dim() != 2 or x.shape[1] != self.d or n < self.nlist:
            raise ValueError(f"need a [n >= nlist, {self.d}] float32 matrix")
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        c = x[torch.randperm(n, generator=g, device=dev)[:self.nlist]].clone()
        for it in range(iters + 1):
            cents = HipFlatIndex(self.d, self.metric, device=self.device)
            cents.add_device(c.contiguous())
            assign = self._assign(cents, x)
            if it == iters:
                break
            sums = torch.zeros((self.nlist, self.d), dtype=torch.float32, device=dev).index_add_(0, assign, x)
            cnt = torch.bincount(assign, minlength=self.nlist).to(torch.float32)
            new = sums / cnt.clamp(min=1.0)[:, None]
            if self.metric == nat.METRIC_IP:                         # spherical k-means: the assignment maximises <x, c>
                new = new / new.norm(dim=1, keepdim=True).clamp(min=1e-20)
            c = torch.where(cnt[:, None] > 0, new, c)                # an empty list keeps its centroid
            cents.close()
        # permute by list; every list on whole 32-row blocks (padding rows: zeros, original id -1)
        order = torch.argsort(assign, stable=True)
        lens = torch.bincount(assign, minlength=self.nlist)
        padded = (lens + 31) // 32 * 32
        offs = torch.zeros(self.nlist + 1, dtype=torch.int64, device=dev)
        offs[1:] = torch.cumsum(padded, 0)
        total = int(offs[-1].item())
        start_unpadded = torch.cumsum(lens, 0) - lens
        pos = offs[:-1][assign[order]] + (torch.arange(n, device=dev) - start_unpadded[assign[order]])
        stored = torch.zeros((total, self.d), dtype=torch.float32, device=dev)
        stored[pos] = x[order]
        orig = torch.full((total,), -1, dtype=torch.int64, device=dev)
        orig[pos] = order
        rows = HipFlatIndex(self.d, self.metric, device=self.device)
        rows.reserve_rows(total)
        step = 1 << 18
        for o in range(0, total, step):
            rows.add_device(stored[o:o + step])
        offs_h = np.ascontiguousarray(offs.cpu().numpy(), dtype=np.int64)
        orig_h = np.ascontiguousarray(orig.cpu().numpy(), dtype=np.int64)
        h = ctypes.c_uint64()
        nat.call("hipivf_create", rows._h, cents._h, offs_h.ctypes.data, orig_h.ctypes.data, self.nlist, ctypes.byref(h))
        self.close()
        self.rows, self.centroids, self._h, self.ntotal = rows, cents, h.value, n
        self.list_lengths = lens.cpu().numpy()

    # ---- search -----------------------------------------------------------------------------------------------------
    def search_device(self, q, k: int, nprobe: int, out=None):
        """q: float32 CUDA tensor [nq, d] -> (scores64, scores32, ids) CUDA tensors [nq, k]; enqueued on the current stream."""
        import torch
        if self._h is None:
            raise RuntimeError("index not built")
        nq = q.shape[0]
        if out is None:
            out = (torch.empty((nq, k), dtype=torch.float64, device=q.device), torch.empty((nq, k), dtype=torch.float32, device=q.device),
                   torch.empty((nq, k), dtype=torch.int64, device=q.device))
        nat.call("hipivf_search_dev", self._h, q.data_ptr(), nq, int(k), int(nprobe), out[0].data_ptr(), out[1].data_ptr(),
                 out[2].data_ptr(), _stream_ptr())
        return out

    def search(self, q, k: int, nprobe: int):
        import torch
        qd = torch.from_numpy(np.ascontiguousarray(np.atleast_2d(q), dtype=np.float32)).to(torch.device("cuda", self.device))
        _, s32, ids = self.search_device(qd, k, nprobe)
        torch.cuda.synchronize()
        return s32.cpu().numpy(), ids.cpu().numpy()

    def close(self) -> None:
        if self._h is not None:
            try:
                nat.call("hipivf_destroy", self._h)
            finally:
                self._h = None
        for ix in (self.rows, self.centroids):
            if ix is not None:
                ix.close()
        self.rows = self.centroids = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
