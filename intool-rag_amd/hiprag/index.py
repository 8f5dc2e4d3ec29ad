"""
HipFlatIndex -- Python handle on libhiprag's dense flat index (the object that stands where the reference
keeps a faiss.IndexFlatL2: rag/storage/faiss_index.py:123-124, searched at :83).

Host arrays are numpy (C-contiguous float32).  The *_device methods take / return torch CUDA tensors and enqueue
on torch's current stream; torch is only the allocator and stream owner here.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np

from . import _native as nat
from ._native import METRIC_IP, METRIC_L2, HipRagError

_METRICS = {"ip": METRIC_IP, "l2": METRIC_L2, METRIC_IP: METRIC_IP, METRIC_L2: METRIC_L2}


def _host_f32(a, cols: Optional[int] = None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim == 1:
        a = a[None, :]
    if a.ndim != 2 or (cols is not None and a.shape[1] != cols):
        raise ValueError(f"expected a [n,{cols}] float32 array, got shape {a.shape}")
    return a


class HipFlatIndex:
    def __init__(self, d: int, metric="l2", device: int = 0, _handle: Optional[int] = None):
        self.device = int(device)
        if _handle is None:
            h = ctypes.c_uint64()
            nat.call("hipidx_create", int(d), _METRICS[metric], self.device, ctypes.byref(h))
            self._h = h.value
        else:
            self._h = _handle
        self.d = int(d)
        self.metric = _METRICS[metric]

    # ---- lifecycle --------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            try:
                nat.call("hipidx_destroy", self._h)
            finally:
                self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- properties (faiss names: ntotal, d) -------------------------------------------------------
    @property
    def ntotal(self) -> int:
        n = ctypes.c_int64()
        nat.call("hipidx_ntotal", self._h, ctypes.byref(n))
        return n.value

    def set_id_base(self, base: int) -> None:
        nat.call("hipidx_set_id_base", self._h, int(base))

    # ---- build -------------------------------------------------------------------------------------
    def add(self, x) -> None:
        """index.add(float32[n,d]) -- rows get ids ntotal..ntotal+n-1 in insertion order."""
        if _is_cuda_tensor(x):
            return self.add_device(x)
        x = _host_f32(x, self.d)
        nat.call("hipidx_add", self._h, x.ctypes.data, x.shape[0])

    def add_device(self, x) -> None:
        import torch
        if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != self.d or x.device.index != self.device:
            raise ValueError("add_device expects a float32 [n,d] tensor on this index's device")
        x = x.contiguous()
        nat.call("hipidx_add_dev", self._h, x.data_ptr(), x.shape[0], _stream_ptr())
        torch.cuda.current_stream().synchronize()   # x may be freed by the caller right after

    # ---- search ------------------------------------------------------------------------------------
    def search(self, q, k: int) -> Tuple[np.ndarray, np.ndarray]:
        """(scores float32 [nq,k], ids int64 [nq,k]) in faiss order; padded with id -1."""
        q = _host_f32(q, self.d)
        nq = q.shape[0]
        scores = np.empty((nq, k), dtype=np.float32)
        ids = np.empty((nq, k), dtype=np.int64)
        nat.call("hipidx_search", self._h, q.ctypes.data, nq, int(k), scores.ctypes.data, ids.ctypes.data)
        return scores, ids

    def search_device(self, q, k: int, out=None):
        """q: float32 CUDA tensor [nq,d].  Returns (scores64 [nq,k] f64, scores [nq,k] f32, ids [nq,k] i64) CUDA
        tensors; work is enqueued on torch's current stream, no synchronisation."""
        import torch
        nq = q.shape[0]
        if out is None:
            dev = q.device
            out = (torch.empty((nq, k), dtype=torch.float64, device=dev),
                   torch.empty((nq, k), dtype=torch.float32, device=dev),
                   torch.empty((nq, k), dtype=torch.int64, device=dev))
        s64, s32, ids = out
        nat.call("hipidx_search_dev", self._h, q.data_ptr(), nq, int(k), s64.data_ptr(), s32.data_ptr(), ids.data_ptr(),
                 _stream_ptr())
        return s64, s32, ids

    def search_begin(self, q, k: int, slot: int = 0, stream: Optional[int] = None) -> None:
        """Phase 1 of a pass (<= 32 queries): enqueue query fragments + the index scan into workspace `slot`.
        `stream` = raw hipStream_t (int) or None for torch's current stream."""
        nat.call("hipidx_search_begin_dev", self._h, q.data_ptr(), q.shape[0], int(k), int(slot),
                 _stream_ptr() if stream is None else ctypes.c_void_p(stream))

    def search_finish(self, q, k: int, slot: int, out, stream: Optional[int] = None):
        """Phase 2: selection, fp64 re-score, top-k, certificate; `out` = (scores64, scores32, ids) CUDA tensors."""
        s64, s32, ids = out
        nat.call("hipidx_search_finish_dev", self._h, q.data_ptr(), q.shape[0], int(k), int(slot), s64.data_ptr(),
                 s32.data_ptr() if s32 is not None else None, ids.data_ptr(),
                 _stream_ptr() if stream is None else ctypes.c_void_p(stream))
        return out

    @property
    def pass_queries(self) -> int:
        """Queries that share one read of the index: 64 by default, 32 in the split / f32 operand modes."""
        n = ctypes.c_int32()
        nat.call("hipidx_pass_queries", self._h, ctypes.byref(n))
        return n.value

    @property
    def launch_queries(self) -> int:
        """Queries one search_begin / search_finish pair takes: the scan runs launch/pass passes inside one launch."""
        n = ctypes.c_int32()
        nat.call("hipidx_launch_queries", self._h, ctypes.byref(n))
        return n.value

    def set_spare_cus(self, n: int) -> None:
        """Leave n CUs out of the scan grid for kernels of other streams (the finish of the previous launch, a BM25 leg, the
        RCCL all-gather); takes effect with the next launch.  See hipidx_set_spare_cus in hiprag.h."""
        nat.call("hipidx_set_spare_cus", self._h, int(n))

    def gate_tail(self, slot: int, stream: int) -> None:
        """hipidx_gate_tail_dev: `stream` waits (in the command processor) until the scan launched after the one of `slot` has
        started -- the finish enqueued behind it then runs on the CUs that scan leaves.  That scan must have been launched."""
        nat.call("hipidx_gate_tail_dev", self._h, int(slot), ctypes.c_void_p(stream))

    @property
    def spare_cus(self) -> int:
        n = ctypes.c_int32()
        nat.call("hipidx_get_spare_cus", self._h, ctypes.byref(n))
        return n.value

    def reserve_search(self, k: int) -> None:
        nat.call("hipidx_reserve_search", self._h, int(k))

    def reserve_rows(self, n_rows: int) -> None:
        """Capacity for n_rows rows in one allocation (hipidx_reserve_rows)."""
        nat.call("hipidx_reserve_rows", self._h, int(n_rows))

    # ---- misc --------------------------------------------------------------------------------------
    def reconstruct(self, row: int) -> np.ndarray:
        out = np.empty(self.d, dtype=np.float32)
        nat.call("hipidx_reconstruct", self._h, int(row), out.ctypes.data)
        return out

    def save(self, path: str) -> None:
        nat.call("hipidx_save", self._h, str(path).encode())

    @classmethod
    def load(cls, path: str, device: int = 0) -> "HipFlatIndex":
        h = ctypes.c_uint64()
        nat.call("hipidx_load", str(path).encode(), int(device), ctypes.byref(h))
        d, m = ctypes.c_int32(), ctypes.c_int32()
        nat.call("hipidx_dim", h.value, ctypes.byref(d))
        nat.call("hipidx_metric", h.value, ctypes.byref(m))
        return cls(d.value, m.value, device, _handle=h.value)

    def enable_timing(self, on=True) -> None:
        """True / 1: HIP events around every scan launch; n > 1: around every n-th launch; False / 0: off."""
        nat.call("hipidx_enable_timing", self._h, int(on))

    def stats(self) -> dict:
        st = nat.HipIdxStats()
        nat.call("hipidx_get_stats", self._h, ctypes.byref(st))
        return {f: getattr(st, f) for f, _ in st._fields_}


def _is_cuda_tensor(x) -> bool:
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


def _stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def merge_topk_device(scores64, ids, k_out: int, metric, out=None):
    """Merge [n_parts, nq, k_in] partial lists (CUDA tensors) -> (scores64, scores32, ids) [nq,k_out].

    The two inputs may be strided views along dim 0 (e.g. halves of one all-gathered [G,2,nq,k] buffer) as long
    as each [nq,k_in] part is contiguous and both use the same part stride."""
    import torch
    n_parts, nq, k_in = scores64.shape
    for t in (scores64, ids):
        if t.stride(2) != 1 or t.stride(1) != k_in:
            raise ValueError("each [nq,k_in] part must be contiguous")
    part_stride = scores64.stride(0) if n_parts > 1 else nq * k_in
    if n_parts > 1 and ids.stride(0) != part_stride:
        raise ValueError("scores and ids must use the same part stride")
    dev = scores64.device
    if out is None:
        out = (torch.empty((nq, k_out), dtype=torch.float64, device=dev),
               torch.empty((nq, k_out), dtype=torch.float32, device=dev),
               torch.empty((nq, k_out), dtype=torch.int64, device=dev))
    nat.call("hiprag_merge_topk_dev", scores64.data_ptr(), ids.data_ptr(), n_parts, nq, k_in, int(k_out),
             int(part_stride), _METRICS[metric], out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), _stream_ptr())
    return out


__all__ = ["HipFlatIndex", "merge_topk_device", "HipRagError", "METRIC_IP", "METRIC_L2"]
