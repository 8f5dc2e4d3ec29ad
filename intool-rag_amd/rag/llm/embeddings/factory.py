"""
Process-wide provider singleton, selected by EMBEDDING_PROVIDER like the reference's factory
(rag/llm/embeddings/factory.py:10-50).  This overlay adds the value "hip"; the reference's other branches
(gemini / hf / ollama) stay where they are in the reference tree -- INTEGRATION.md shows the three lines to add there.
"""
import os
import threading
from typing import Optional

from rag.llm.embeddings.base import EmbeddingProvider

_PROVIDER: Optional[EmbeddingProvider] = None
_LOCK = threading.Lock()     # embed calls arrive from asyncio.to_thread workers; the reference's global is unlocked


def get_embedding_provider() -> EmbeddingProvider:
    global _PROVIDER
    if _PROVIDER is not None:
        return _PROVIDER
    with _LOCK:
        if _PROVIDER is None:
            name = os.getenv("EMBEDDING_PROVIDER", "hip").lower()
            if name != "hip":
                raise RuntimeError(f"EMBEDDING_PROVIDER={name!r}: this overlay only ships the 'hip' provider; "
                                   "the reference tree provides gemini / hf / ollama")
            from rag.providers.hip.embeddings import HipEmbeddingProvider
            _PROVIDER = HipEmbeddingProvider()
    return _PROVIDER


def set_embedding_provider(provider: Optional[EmbeddingProvider]) -> None:
    """Test hook / explicit wiring (the reference has no equivalent; its singleton is never reset)."""
    global _PROVIDER
    _PROVIDER = provider
