"""
Reciprocal-rank fusion on the GPU (csrc/merge_rrf.hip).  The reference names hybrid fusion (README.md:54-58) and
carries weights BM25_WEIGHT=0.3 / VECTOR_WEIGHT=0.7 (rag/config.py:44-45) that no code reads; spec in DESIGN.md:
s(d) = w_a/(c + rank_a) + w_b/(c + rank_b), c = 60, 1-based ranks, a missing list adds 0, fp32, (s desc, id asc).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from . import _native as nat

RRF_C = 60.0


def rrf_fuse(ids_a, ids_b, k: int, c: float = RRF_C, w_a: float = 1.0, w_b: float = 1.0) -> Tuple[np.ndarray, np.ndarray]:
    a = np.ascontiguousarray(np.atleast_2d(ids_a), dtype=np.int64)
    b = np.ascontiguousarray(np.atleast_2d(ids_b), dtype=np.int64)
    if a.shape[0] != b.shape[0]:
        raise ValueError("both lists need the same number of queries")
    nq = a.shape[0]
    scores = np.empty((nq, k), dtype=np.float32)
    ids = np.empty((nq, k), dtype=np.int64)
    nat.call("hiprrf_fuse", a.ctypes.data, b.ctypes.data, nq, a.shape[1], b.shape[1], int(k), float(c), float(w_a),
             float(w_b), scores.ctypes.data, ids.ctypes.data)
    return scores, ids


def rrf_fuse_device(ids_a, ids_b, k: int, c: float = RRF_C, w_a: float = 1.0, w_b: float = 1.0, out=None):
    """ids_a/ids_b: int64 CUDA tensors [nq, depth].  Enqueues on torch's current stream."""
    import torch
    from .index import _stream_ptr
    nq = ids_a.shape[0]
    if out is None:
        out = (torch.empty((nq, k), dtype=torch.float32, device=ids_a.device),
               torch.empty((nq, k), dtype=torch.int64, device=ids_a.device))
    nat.call("hiprrf_fuse_dev", ids_a.data_ptr(), ids_b.data_ptr(), nq, ids_a.shape[1], ids_b.shape[1], int(k), float(c),
             float(w_a), float(w_b), out[0].data_ptr(), out[1].data_ptr(), _stream_ptr())
    return out
