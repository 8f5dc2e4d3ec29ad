"""
BM25 side index over the same chunk table the dense index uses (row id == position in `{doc_id}_chunks.json`,
rag/storage/faiss_index.py:175-181), built once per chunk-file version and kept in HBM (hiprag.HipBM25).
The reference names this leg (README.md:54-58, rag/config.py:43-45) without implementing it; spec in DESIGN.md.
"""
from __future__ import annotations

import threading
from pathlib import Path
from typing import Any, Dict, List, Tuple

from rag.config import config

_SPARSE_CACHE: Dict[str, Tuple[int, Any]] = {}
_LOCK = threading.Lock()


def get_sparse_index(storage_path: Path, doc_id: str, chunks_list: List[Dict[str, Any]]):
    from hiprag import HipBM25, build_postings_from_texts
    key = str(Path(storage_path) / doc_id)
    with _LOCK:
        hit = _SPARSE_CACHE.get(key)
        if hit is not None and hit[0] == id(chunks_list):
            return hit[1]
        if hit is not None and isinstance(hit[0], tuple):          # built at ingest time from the same texts
            texts = [c.get("text", "") for c in chunks_list]
            if hit[0] == ("texts", len(texts), hash(tuple(texts))):
                _SPARSE_CACHE[key] = (id(chunks_list), hit[1])
                return hit[1]
    postings = build_postings_from_texts([c.get("text", "") for c in chunks_list])
    index = HipBM25(postings, device=config.HIP_DEVICE)
    with _LOCK:
        _SPARSE_CACHE[key] = (id(chunks_list), index)
    return index


def put_sparse_index(storage_path: Path, doc_id: str, texts: List[str]) -> int:
    """Ingest side: build the postings of `texts` (row id == position) and keep the index for the readers of this
    document; get_sparse_index rebuilds only if it is later handed a different chunk table.  Returns the postings count."""
    from hiprag import HipBM25, build_postings_from_texts
    postings = build_postings_from_texts(texts)
    index = HipBM25(postings, device=config.HIP_DEVICE)
    with _LOCK:
        _SPARSE_CACHE[str(Path(storage_path) / doc_id)] = (("texts", len(texts), hash(tuple(texts))), index)
    return int(postings.offsets[-1])


def clear_sparse_cache() -> None:
    with _LOCK:
        _SPARSE_CACHE.clear()
