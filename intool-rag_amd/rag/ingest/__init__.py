"""Ingest-side counterpart of the retrieval hot path: phase 4 (embeddings + indexing) of the reference's
IngestionPipeline (rag/ingest/ingestion_pipeline.py:80-94).  The earlier phases (OCR, semantic tree, chunking) are the
reference's own and out of scope."""
from rag.ingest.indexing import index_chunks

__all__ = ["index_chunks"]
