"""
Import / export of FAISS flat index files, so existing `{doc_id}_faiss.index` files written by the reference
(`faiss.write_index`, rag/storage/faiss_index.py:133) can be loaded into a HipFlatIndex without re-embedding
(SURVEY.md 8f, row f2).

Format of IndexFlatL2 / IndexFlatIP in faiss 1.7.x (index_write.cpp, restated from the published source; faiss is not
installed in this image, so this reader/writer pair is tested against itself and against hand-built bytes only --
NOT yet against a file produced by faiss):

    u32  fourcc            "IxF2" (L2) or "IxFI" (inner product), little endian
    i32  d
    i64  ntotal
    i64  dummy, i64 dummy  (1 << 20 each)
    u8   is_trained
    i32  metric_type       0 = inner product, 1 = L2
    u64  n                 number of 4-byte words that follow (= ntotal * d)
    f32  data[n]           row-major vectors
"""
from __future__ import annotations

import struct
from typing import Tuple

import numpy as np

_FOURCC = {b"IxF2": 1, b"IxFI": 0}          # -> hiprag metric (1 = L2, 0 = IP)
_HEADER = struct.Struct("<4siqqqBi")


def read_faiss_flat(path: str) -> Tuple[int, np.ndarray]:
    """-> (metric, float32 [ntotal, d]).  Raises ValueError for anything that is not a flat L2 / IP index."""
    with open(path, "rb") as f:
        head = f.read(_HEADER.size)
        if len(head) != _HEADER.size:
            raise ValueError(f"{path}: too short for a FAISS index header")
        fourcc, d, ntotal, _d1, _d2, _trained, metric_type = _HEADER.unpack(head)
        if fourcc not in _FOURCC:
            raise ValueError(f"{path}: fourcc {fourcc!r} is not a flat index (IxF2 / IxFI)")
        if d <= 0 or ntotal < 0:
            raise ValueError(f"{path}: bad header (d={d}, ntotal={ntotal})")
        (nwords,) = struct.unpack("<Q", f.read(8))
        if nwords != ntotal * d:
            raise ValueError(f"{path}: payload of {nwords} words does not match ntotal*d = {ntotal * d}")
        data = np.fromfile(f, dtype="<f4", count=nwords)
        if data.size != nwords:
            raise ValueError(f"{path}: truncated payload")
    metric = _FOURCC[fourcc]
    if metric_type not in (0, 1) or (metric_type == 1) != (metric == 1):
        raise ValueError(f"{path}: metric_type {metric_type} inconsistent with fourcc {fourcc!r}")
    return metric, data.reshape(ntotal, d)


def write_faiss_flat(path: str, x: np.ndarray, metric: int) -> None:
    """Write rows as an IndexFlatL2 (metric 1) / IndexFlatIP (metric 0) file."""
    x = np.ascontiguousarray(x, dtype="<f4")
    if x.ndim != 2:
        raise ValueError("x must be [n, d]")
    fourcc = b"IxF2" if metric == 1 else b"IxFI"
    with open(path, "wb") as f:
        f.write(_HEADER.pack(fourcc, x.shape[1], x.shape[0], 1 << 20, 1 << 20, 1, 1 if metric == 1 else 0))
        f.write(struct.pack("<Q", x.size))
        x.tofile(f)


def load_faiss_flat_into_hip(path: str, device: int = 0):
    """FAISS flat file -> HipFlatIndex with the same metric and row order."""
    from .index import HipFlatIndex
    metric, x = read_faiss_flat(path)
    index = HipFlatIndex(int(x.shape[1]), metric, device=device)
    index.add(x)
    return index
