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


def hybrid_search(index, bm25, queries, sparse_queries, depth: int = 50, k: int = 10, c: float = 60.0,
                  w_dense: float = 1.0, w_sparse: float = 1.0) -> Tuple[np.ndarray, np.ndarray]:
    """hiphybrid_search: dense top-`depth` + BM25 top-`depth` + RRF -> top-k in ONE library call (host arrays in and out,
    the two intermediate lists stay on the GPU).  `index` is a HipFlatIndex, `bm25` a HipBM25 on the same device,
    `queries` [nq, d] float32, `sparse_queries` nq lists of term ids.  Returns (scores float32 [nq, k], ids int64 [nq, k])."""
    q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32))
    if q.ndim != 2 or q.shape[1] != index.d or len(sparse_queries) != q.shape[0]:
        raise ValueError("queries must be [nq, d] with one term list per query")
    terms, qoff = bm25._flatten(sparse_queries)
    nq = q.shape[0]
    scores = np.empty((nq, k), dtype=np.float32)
    ids = np.empty((nq, k), dtype=np.int64)
    nat.call("hiphybrid_search", index._h, bm25._h, q.ctypes.data, terms.ctypes.data if terms.size else None,
             qoff.ctypes.data, nq, int(depth), int(k), float(c), float(w_dense), float(w_sparse), scores.ctypes.data,
             ids.ctypes.data)
    return scores, ids


def hybrid_search_device(index, bm25, q_dev, sparse_queries, depth: int = 50, k: int = 10, c: float = 60.0,
                         w_dense: float = 1.0, w_sparse: float = 1.0, return_lists: bool = False):
    """hiphybrid_search_dev, the device-resident form of hybrid_search: q_dev is a float32 CUDA tensor [nq, d]; returns
    (fused scores float32 [nq, k], ids int64 [nq, k]) as CUDA tensors ordered on the current stream.  The library runs the
    two legs beside each other (dense on its high-priority stream with CUs left out of the scan grid, BM25 on a helper
    stream that fills them) and the fusion on the current stream behind both.  return_lists=True adds the two per-leg
    lists: ((dense scores float64, dense ids), (BM25 scores float64, BM25 ids)), each [nq, depth]."""
    import torch
    from .index import _stream_ptr
    if not (q_dev.is_cuda and q_dev.dtype == torch.float32 and q_dev.dim() == 2 and q_dev.shape[1] == index.d
            and q_dev.is_contiguous()):
        raise ValueError("q_dev must be a contiguous float32 CUDA tensor [nq, d]")
    nq = q_dev.shape[0]
    if len(sparse_queries) != nq:
        raise ValueError("one term list per query")
    terms, qoff = bm25._flatten(sparse_queries)
    dev = q_dev.device
    lists = torch.empty((4, nq, depth), dtype=torch.int64, device=dev)
    out = (torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev))
    if nq:
        nat.call("hiphybrid_search_dev", index._h, bm25._h, q_dev.data_ptr(), terms.ctypes.data if terms.size else None,
                 qoff.ctypes.data, nq, int(depth), int(k), float(c), float(w_dense), float(w_sparse), lists.data_ptr(),
                 out[0].data_ptr(), out[1].data_ptr(), _stream_ptr())
    if return_lists:
        return out + (((lists[0].view(torch.float64), lists[1]), (lists[2].view(torch.float64), lists[3])),)
    return out
