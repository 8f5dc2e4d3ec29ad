"""
The handful of settings the hot path reads, under the SAME environment names as the reference's rag/config.py
(:9-11, :25-30, :41-45, :53-62) so an existing deployment's .env keeps working.  Unlike the reference this module
has no import-time side effects (the reference mkdirs at import, config.py:50,63,66).
"""
import os
from pathlib import Path


class Config:
    EMBEDDING_MODEL = os.getenv("EMBEDDING_MODEL", "BAAI/bge-m3")
    EMBEDDING_BATCH_SIZE = int(os.getenv("EMBEDDING_BATCH_SIZE", "8"))
    VECTOR_DIMENSION = int(os.getenv("VECTOR_DIMENSION", "1024"))
    RERANKER_MODEL = os.getenv("RERANKER_MODEL", "BAAI/bge-reranker-v2-m3")
    RERANKER_ENABLED = os.getenv("RERANKER_ENABLED", "true").lower() == "true"
    RERANKER_TOP_K = int(os.getenv("RERANKER_TOP_K", "10"))
    RETRIEVAL_TOP_K = int(os.getenv("RETRIEVAL_TOP_K", "10"))
    RETRIEVAL_MIN_SCORE = float(os.getenv("RETRIEVAL_MIN_SCORE", "0.3"))
    HYBRID_SEARCH_ENABLED = os.getenv("HYBRID_SEARCH_ENABLED", "true").lower() == "true"
    BM25_WEIGHT = float(os.getenv("BM25_WEIGHT", "0.3"))
    VECTOR_WEIGHT = float(os.getenv("VECTOR_WEIGHT", "0.7"))
    EMBEDDING_QUERY_INSTRUCTION = os.getenv("EMBEDDING_QUERY_INSTRUCTION",
                                            "Represent this sentence for searching relevant passages: ")
    # additions of this build
    HIP_INDEX_METRIC = os.getenv("HIP_INDEX_METRIC", "l2")       # the reference builds IndexFlatL2 (faiss_index.py:123)
    HIP_DEVICE = int(os.getenv("HIP_DEVICE", "0"))
    HIP_COMPAT_MINUS_ONE = os.getenv("HIP_COMPAT_MINUS_ONE", "true").lower() == "true"
    # false (default): `instruction` arguments are accepted and ignored, exactly like the reference's HF provider
    # (hf/embeddings.py:45,64-65); true: a given instruction is prepended to each text before tokenisation (BGE recipe)
    HIP_APPLY_INSTRUCTION = os.getenv("HIP_APPLY_INSTRUCTION", "false").lower() == "true"
    EMBEDDING_PASSAGE_INSTRUCTION = os.getenv("EMBEDDING_PASSAGE_INSTRUCTION", "")
    # false (default): search the FIRST index file only, like the reference (faiss_index.py:162-167); true: every document
    HIP_SEARCH_ALL_DOCUMENTS = os.getenv("HIP_SEARCH_ALL_DOCUMENTS", "false").lower() == "true"

    @property
    def STORAGE_DIR(self) -> Path:
        return Path(os.getenv("STORAGE_DIR", "./storages"))


config = Config()
