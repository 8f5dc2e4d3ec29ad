"""
Drop-in overlay for batd-htplus/intool-rag's `rag` package: only the modules on the retrieval hot path.

Copy `rag/providers/hip/`, `rag/storage/hip_index/` and `rag/query/{retriever,reranker}.py` into the reference tree
(see INTEGRATION.md); the small mirrors here (`rag/config.py`, `rag/logging.py`, `rag/llm/embeddings/*`) exist so this
overlay also runs standalone, and follow the reference's names, arguments and error behaviour.
"""
