"""The provider interface of the reference, rag/llm/embeddings/base.py:5-17, restated so the overlay runs standalone."""
import abc
from typing import List, Optional


class EmbeddingProvider(abc.ABC):
    """Three methods; both embed calls are coroutines and return plain Python lists of floats."""

    @abc.abstractmethod
    async def embed_single(self, text: str, instruction: Optional[str] = None) -> List[float]:
        ...

    @abc.abstractmethod
    async def embed_batch(self, texts: List[str], instruction: Optional[str] = None) -> List[List[float]]:
        ...

    @abc.abstractmethod
    def dimension(self) -> int:
        ...
