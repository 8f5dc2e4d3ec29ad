"""
rag/ingest/indexing.py -- phase 4 of IngestionPipeline.ingest_pdf (rag/ingest/ingestion_pipeline.py:80-94) on the GPU.

Reference:   texts = [c.text for c in chunks]
             embeddings = await embedding_provider.embed_batch(texts)          # list of lists of Python floats
             index = create_faiss_index(embeddings)                            # np.array(..., float32) -> IndexFlatL2.add
             save_faiss_index(index, STORAGE_DIR / f"{doc_id}_faiss.index")
Here the vectors never leave HBM: the provider's encoder output (a CUDA tensor) is re-tiled straight into the index
(hipidx_add_dev), the index file is written once, and -- when hybrid search is on -- the BM25 postings of the same
chunk texts are built and uploaded in the same call, so the first query does not pay for them.
Any EmbeddingProvider works; providers without a device path go through the reference's list-of-lists interface.
Row id == position in `chunks`, the convention the reader relies on (rag/storage/faiss_index.py:175-181).
"""
from __future__ import annotations

import time
from pathlib import Path
from typing import Any, Dict, Optional, Sequence

from rag.config import config
from rag.llm.embeddings.factory import get_embedding_provider
from rag.logging import logger
from rag.storage.hip_index import INDEX_SUFFIX, create_hip_index, save_hip_index


def _text_of(chunk: Any) -> str:
    return chunk["text"] if isinstance(chunk, dict) else chunk.text


async def index_chunks(doc_id: str, chunks: Sequence[Any], storage_dir: Optional[Path] = None, provider=None,
                       with_sparse: Optional[bool] = None) -> Dict[str, Any]:
    """Embed `chunks` (objects with .text, or dicts with "text"), build and save `{doc_id}_hip.index`.
    Returns the same summary keys the reference's ingest returns for this phase."""
    start = time.time()
    storage = Path(storage_dir) if storage_dir is not None else config.STORAGE_DIR
    provider = provider or get_embedding_provider()
    texts = [_text_of(c) for c in chunks]
    logger.info(f"Generating embeddings for {len(texts)} chunks...")
    if hasattr(provider, "embed_batch_device"):
        embeddings = await provider.embed_batch_device(texts)        # CUDA tensor [n, d]
    else:
        embeddings = await provider.embed_batch(texts)
    if len(texts) == 0:
        raise ValueError("index_chunks needs at least one chunk")
    index = create_hip_index(embeddings)
    index_path = storage / f"{doc_id}{INDEX_SUFFIX}"
    save_hip_index(index, str(index_path))
    sparse = config.HYBRID_SEARCH_ENABLED if with_sparse is None else with_sparse
    postings = 0
    if sparse:
        from rag.storage.hip_index.sparse import put_sparse_index
        postings = put_sparse_index(storage, doc_id, texts)
    total = time.time() - start
    logger.info(f"Indexing complete in {total:.2f}s")
    return {"success": True, "doc_id": doc_id, "chunk_count": len(texts), "vectors_indexed": int(index.ntotal),
            "postings_indexed": postings, "index_path": str(index_path), "processing_time": total}
