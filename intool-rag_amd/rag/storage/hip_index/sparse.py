"""
BM25 side index over the same chunk table the dense index uses (row id == position in `{doc_id}_chunks.json`,
rag/storage/faiss_index.py:175-181), built once per chunk-file version and kept in HBM (hiprag.HipBM25).
The reference names this leg (README.md:54-58, rag/config.py:43-45) without implementing it; spec in DESIGN.md.
"""
from __future__ import annotations

import hashlib
import threading
from pathlib import Path
from typing import Any, Dict, List, Tuple

from rag.config import config

_SPARSE_CACHE: Dict[str, Tuple[tuple, Any]] = {}
_LOCK = threading.Lock()


def _table_version(storage_path: Path, doc_id: str, n_rows: int) -> tuple:
    """Version key of a document's chunk table: the one `_load_chunk_list` caches by -- (path, mtime) -- plus the row
    count.  (An object identity such as id(chunks_list) is NOT a version: once a reloaded table frees the old list,
    CPython may hand its id to the new one and the old postings would be searched against the new rows.)"""
    path = Path(storage_path) / f"{doc_id}_chunks.json"
    try:
        return ("file", str(path), path.stat().st_mtime, n_rows)
    except OSError:
        return ("file", str(path), None, n_rows)


def _digest(texts: List[str]) -> bytes:
    h = hashlib.blake2b(digest_size=16)
    for t in texts:
        h.update(t.encode("utf-8", "surrogatepass"))
        h.update(b"\x00")
    return h.digest()


def get_sparse_index(storage_path: Path, doc_id: str, chunks_list: List[Dict[str, Any]]):
    from hiprag import HipBM25, build_postings_from_texts
    key = str(Path(storage_path) / doc_id)
    version = _table_version(storage_path, doc_id, len(chunks_list))
    with _LOCK:
        hit = _SPARSE_CACHE.get(key)
        if hit is not None and hit[0] == version:
            return hit[1]
    texts = [c.get("text", "") for c in chunks_list]
    if hit is not None and hit[0][0] == "ingest" and hit[0][1:] == (len(texts), _digest(texts)):
        with _LOCK:                                         # built at ingest time from these very texts: adopt it
            _SPARSE_CACHE[key] = (version, hit[1])
        return hit[1]
    index = HipBM25(build_postings_from_texts(texts), device=config.HIP_DEVICE)
    with _LOCK:
        _SPARSE_CACHE[key] = (version, index)
    return index


def put_sparse_index(storage_path: Path, doc_id: str, texts: List[str]) -> int:
    """Ingest side: build the postings of `texts` (row id == position) and keep the index for the readers of this
    document; get_sparse_index adopts it for the chunk-file version whose texts have the same digest and rebuilds
    for any other.  Returns the postings count."""
    from hiprag import HipBM25, build_postings_from_texts
    postings = build_postings_from_texts(texts)
    index = HipBM25(postings, device=config.HIP_DEVICE)
    with _LOCK:
        _SPARSE_CACHE[str(Path(storage_path) / doc_id)] = (("ingest", len(texts), _digest(list(texts))), index)
    return int(postings.offsets[-1])


def clear_sparse_cache() -> None:
    with _LOCK:
        _SPARSE_CACHE.clear()
