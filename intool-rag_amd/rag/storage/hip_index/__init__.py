"""
rag/storage/hip_index -- MI355X drop-in for the reference's rag/storage/faiss_index.py.

Same function surface, same argument meaning, same return shapes and error behaviour:

    reference (rag/storage/faiss_index.py)            this module
    ----------------------------------------------    -------------------------------------------
    FAISSIndexReader(index_path)            :26-61    HipIndexReader(index_path)
      .search(query_embedding, top_k=10)    :63-91      .search(...)  -> [(id, score)], score = clamp(1 - d/2, 0, 1)
      .get_dimension() / .get_size()        :93-103     same
    create_faiss_index(embeddings)          :106-128  create_hip_index(embeddings)
    save_faiss_index(index, path)           :131-134  save_hip_index(index, path)
    async search_faiss_by_vector(vec, limit=50, project=None)  :137-199   async search_hip_by_vector(...)
    async initialize_storage()              :202-228  async initialize_storage()
    _INDEX_CACHE                            :24       _INDEX_CACHE (path-keyed, process lifetime)

The arithmetic runs in libhiprag.so (HipFlatIndex); there is no CPU fallback: without the library or a GPU these
functions raise RuntimeError exactly where the reference raises "FAISS not installed" (:36-37, :118-119).

Differences, all deliberate and listed in DESIGN.md:
  * index files are `{doc_id}_hip.index` (HIPIDX01: header + row-major fp32), not FAISS's binary format;
  * the chunk table (`{doc_id}_chunks.json`) is parsed once per file version and cached -- the reference re-parses it on
    every query (:172-176);
  * HIP_SEARCH_ALL_DOCUMENTS=true searches every document's index and merges (the reference, and the default here,
    search only the first file, :162-167);
  * HIP_COMPAT_MINUS_ONE=true (default) keeps the reference's quirk that an id of -1 (k > ntotal) passes
    `faiss_id < len(chunks)` and indexes the LAST chunk with score 0 (:179-181); set it to false to drop such rows.
"""
from __future__ import annotations

import json
import threading
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from rag.config import config
from rag.logging import logger

try:
    from hiprag import HipFlatIndex, HipRagError
    from hiprag import _native as _nat
    _nat.load()
    HAS_HIP = True
    _HIP_IMPORT_ERROR: Optional[Exception] = None
except Exception as _e:  # library not built / not on sys.path
    HAS_HIP = False
    _HIP_IMPORT_ERROR = _e

INDEX_SUFFIX = "_hip.index"
FAISS_SUFFIX = "_faiss.index"      # files the reference wrote; imported on load (hiprag/faiss_io.py)
_INDEX_CACHE: Dict[str, "HipFlatIndex"] = {}
_CHUNK_CACHE: Dict[str, Tuple[float, List[Dict[str, Any]]]] = {}
_LOCK = threading.Lock()


def _require_hip() -> None:
    if not HAS_HIP:
        raise RuntimeError(f"libhiprag not available (build intool-rag_amd/lib/libhiprag.so): {_HIP_IMPORT_ERROR}")


class HipIndexReader:
    """Read-only index wrapper with the reference's caching behaviour (faiss_index.py:26-103)."""

    def __init__(self, index_path: str):
        _require_hip()
        self.index_path = str(index_path)
        self.index: Optional[HipFlatIndex] = None
        self._load_index()

    def _load_index(self) -> None:
        with _LOCK:
            cached = _INDEX_CACHE.get(self.index_path)
            if cached is not None:
                self.index = cached
                return
            try:
                if self.index_path.endswith(FAISS_SUFFIX):
                    # an index written by the reference's faiss.write_index (faiss_index.py:133): import it as is
                    from hiprag.faiss_io import load_faiss_flat_into_hip
                    self.index = load_faiss_flat_into_hip(self.index_path, device=config.HIP_DEVICE)
                else:
                    self.index = HipFlatIndex.load(self.index_path, device=config.HIP_DEVICE)
            except Exception as e:
                raise RuntimeError(f"Failed to load HIP index: {e}")
            _INDEX_CACHE[self.index_path] = self.index
        logger.info(f"Loaded HIP index: {self.index_path}")
        logger.info(f"  Dimension: {self.index.d}")
        logger.info(f"  Size: {self.index.ntotal} vectors")

    def search(self, query_embedding: List[float], top_k: int = 10) -> List[Tuple[int, float]]:
        """[(embedding_id, score)], ids of -1 included, in the index's order (ascending distance for L2)."""
        if self.index is None:
            raise RuntimeError("Index not loaded")
        query_np = np.array([query_embedding], dtype=np.float32)
        values, indices = self.index.search(query_np, top_k)
        results = []
        l2 = self.index.metric == 1
        for idx, val in zip(indices[0], values[0]):
            val = float(val)                       # the reference's numpy<2 promotes float32 scalars to float64 here
            score = 1.0 - (val / 2.0) if l2 else val   # inner product of unit vectors IS 1 - d/2
            score = max(0.0, min(1.0, score))
            results.append((int(idx), float(score)))
        return results

    def search_raw(self, query_embedding: List[float], top_k: int = 10) -> List[Tuple[int, float]]:
        """Raw distances / inner products (the agent path derives 1/(1+d) from them, rag/agent/search_engine.py:50)."""
        if self.index is None:
            raise RuntimeError("Index not loaded")
        values, indices = self.index.search(np.array([query_embedding], dtype=np.float32), top_k)
        return [(int(i), float(v)) for i, v in zip(indices[0], values[0])]

    def get_dimension(self) -> int:
        if self.index is None:
            raise RuntimeError("Index not loaded")
        return self.index.d

    def get_size(self) -> int:
        if self.index is None:
            raise RuntimeError("Index not loaded")
        return self.index.ntotal


def create_hip_index(embeddings, metric: Optional[str] = None) -> "HipFlatIndex":
    """Build an index from a list of vectors (ingest phase, rag/ingest/ingestion_pipeline.py:88).

    Accepts the reference's list-of-lists as well as a float32 ndarray or a CUDA tensor (no list round trip)."""
    _require_hip()
    if hasattr(embeddings, "is_cuda") and embeddings.is_cuda:
        d = int(embeddings.shape[1])
        index = HipFlatIndex(d, metric or config.HIP_INDEX_METRIC, device=config.HIP_DEVICE)
        index.add_device(embeddings)
    else:
        embeddings_np = np.asarray(embeddings, dtype=np.float32)
        if embeddings_np.ndim != 2:
            raise ValueError(f"embeddings must be [n, d], got shape {embeddings_np.shape}")
        index = HipFlatIndex(int(embeddings_np.shape[1]), metric or config.HIP_INDEX_METRIC, device=config.HIP_DEVICE)
        index.add(embeddings_np)
    logger.info(f"Created HIP index: {index.ntotal} vectors, dim={index.d}")
    return index


def save_hip_index(index: "HipFlatIndex", path: str) -> None:
    index.save(str(path))
    with _LOCK:
        _INDEX_CACHE[str(path)] = index          # the freshly built index is what readers of this path must see
    logger.info(f"Saved HIP index to {path}")


def _load_chunk_list(storage_dir: Path, doc_id: str) -> List[Dict[str, Any]]:
    """`list(load_chunks(doc_id).values())` of the reference (file_storage.py:139-166, faiss_index.py:175-176):
    chunks in file order, later duplicates of a chunk_id REPLACING the earlier dict entry in place."""
    path = storage_dir / f"{doc_id}_chunks.json"
    if not path.exists():
        raise FileNotFoundError(f"Chunks not found: {path}")
    mtime = path.stat().st_mtime
    key = str(path)
    with _LOCK:
        hit = _CHUNK_CACHE.get(key)
        if hit is not None and hit[0] == mtime:
            return hit[1]
    with open(path, "r", encoding="utf-8") as f:
        data = json.load(f)
    by_id: Dict[str, Dict[str, Any]] = {}
    for chunk in data.get("chunks", []):
        by_id[chunk["chunk_id"]] = chunk
    chunks = list(by_id.values())
    with _LOCK:
        _CHUNK_CACHE[key] = (mtime, chunks)
    return chunks


def enrich(search_results: List[Tuple[int, float]], chunks_list: List[Dict[str, Any]],
           compat_minus_one: Optional[bool] = None) -> List[dict]:
    """faiss_index.py:178-192 -- same dict schema, same `faiss_id < len(list)` admission rule."""
    if compat_minus_one is None:
        compat_minus_one = config.HIP_COMPAT_MINUS_ONE
    enriched = []
    for row_id, score in search_results:
        if row_id < 0 and not compat_minus_one:
            continue
        if row_id < len(chunks_list):
            chunk = chunks_list[row_id]
            meta = chunk.get("metadata", {})
            enriched.append({
                "chunk_id": chunk.get("chunk_id", f"unknown_{row_id}"),
                "text": chunk.get("text", ""),
                "score": score,
                "page": chunk.get("page", 0),
                "chapter": meta.get("chapter"),
                "section": meta.get("section"),
                "subsection": meta.get("subsection"),
                "title": meta.get("title"),
                "source_filename": meta.get("source_filename"),
            })
    return enriched


def open_first_index():
    """(reader, doc_id, chunk list) of the FIRST index file in STORAGE_DIR -- the one the reference searches
    (faiss_index.py:162-176) -- or None when there is no index."""
    storage_path = Path(config.STORAGE_DIR)
    suffix = INDEX_SUFFIX
    index_files = list(storage_path.glob(f"*{INDEX_SUFFIX}"))
    if not index_files:                                   # fall back to indices the reference itself wrote
        suffix = FAISS_SUFFIX
        index_files = list(storage_path.glob(f"*{FAISS_SUFFIX}"))
    if not index_files:
        return None
    index_path = index_files[0]
    doc_id = index_path.name[:-len(suffix)]
    return HipIndexReader(str(index_path)), doc_id, _load_chunk_list(storage_path, doc_id)


def open_all_indices():
    """[(reader, doc_id, chunk list)] of EVERY document under STORAGE_DIR, in file-name order (a `_hip.index` file wins
    over the same document's `_faiss.index`).  Documents whose chunk table is missing are skipped with a warning."""
    storage_path = Path(config.STORAGE_DIR)
    by_doc: Dict[str, Path] = {}
    for suffix in (FAISS_SUFFIX, INDEX_SUFFIX):            # later suffix overrides
        for f in storage_path.glob(f"*{suffix}"):
            by_doc[f.name[:-len(suffix)]] = f
    opened = []
    for doc_id in sorted(by_doc):
        try:
            chunks = _load_chunk_list(storage_path, doc_id)
        except FileNotFoundError as e:
            logger.warning(f"Skipping {doc_id}: {e}")
            continue
        opened.append((HipIndexReader(str(by_doc[doc_id])), doc_id, chunks))
    return opened


def search_all_documents(query_vector: List[float], limit: int = 50) -> List[dict]:
    """Multi-document search (the reference stops at the first file, faiss_index.py:162-167): the top `limit`
    of every document's index, enriched against that document's own chunk table, merged by (score desc, document
    order, rank inside the document) and cut to `limit`.  Scores are the same clamp(1 - d/2) on unit vectors in every
    document, so they compare across documents.  Padding rows (id -1, k > ntotal) never enter the merge."""
    merged = []
    for di, (reader, doc_id, chunks_list) in enumerate(open_all_indices()):
        results = [r for r in reader.search(query_vector, top_k=limit) if r[0] >= 0]
        for rank, row in enumerate(enrich(results, chunks_list, compat_minus_one=False)):
            row["doc_id"] = doc_id
            merged.append((-row["score"], di, rank, row))
    merged.sort(key=lambda t: t[:3])
    return [t[3] for t in merged[:limit]]


async def search_hip_by_vector(query_vector: List[float], limit: int = 50, project: Optional[str] = None) -> List[dict]:
    """Main search function of the query pipeline (faiss_index.py:137-199): first index file in STORAGE_DIR, enriched
    results; `project` is accepted and ignored like in the reference; no index -> [] with a warning.
    HIP_SEARCH_ALL_DOCUMENTS=true searches every document instead (search_all_documents)."""
    try:
        if config.HIP_SEARCH_ALL_DOCUMENTS:
            enriched_results = search_all_documents(query_vector, limit)
            if not enriched_results:
                logger.warning("No HIP indices found")
            logger.info(f"HIP search returned {len(enriched_results)} results")
            return enriched_results
        opened = open_first_index()
        if opened is None:
            logger.warning("No HIP indices found")
            return []
        reader, _doc_id, chunks_list = opened
        search_results = reader.search(query_vector, top_k=limit)
        enriched_results = enrich(search_results, chunks_list)
        logger.info(f"HIP search returned {len(enriched_results)} results")
        return enriched_results
    except Exception as e:
        logger.error(f"HIP search failed: {e}")
        raise


async def initialize_storage() -> None:
    """Pre-load every index under STORAGE_DIR into HBM and warm the chunk tables (faiss_index.py:202-228)."""
    try:
        storage_path = Path(config.STORAGE_DIR)
        if not storage_path.exists():
            logger.warning(f"Storage directory not found: {storage_path}")
            return
        count = 0
        files = [(f, INDEX_SUFFIX) for f in storage_path.glob(f"*{INDEX_SUFFIX}")]
        files += [(f, FAISS_SUFFIX) for f in storage_path.glob(f"*{FAISS_SUFFIX}")]
        for index_file, suffix in files:
            try:
                HipIndexReader(str(index_file))
                try:
                    _load_chunk_list(storage_path, index_file.name[:-len(suffix)])
                except FileNotFoundError:
                    pass
                count += 1
            except Exception as e:
                logger.error(f"Failed to pre-load index {index_file}: {e}")
        logger.info(f"Initialized storage: Loaded {count} indices into HBM")
    except Exception as e:
        logger.error(f"Storage initialization failed: {e}")


def clear_caches() -> None:
    with _LOCK:
        _INDEX_CACHE.clear()
        _CHUNK_CACHE.clear()
    from rag.storage.hip_index.sparse import clear_sparse_cache
    clear_sparse_cache()                 # postings are versioned by the chunk table they were built from


__all__ = ["HipIndexReader", "create_hip_index", "save_hip_index", "search_hip_by_vector", "initialize_storage",
           "enrich", "clear_caches", "open_first_index", "open_all_indices", "search_all_documents", "HAS_HIP",
           "INDEX_SUFFIX"]
