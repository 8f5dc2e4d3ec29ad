"""
Module-level convenience entry points over the provider singleton.

The reference exposes the same three names (rag/llm/embeddings/service.py:5-14): `embed`, `embed_batch`,
`embedding_dim`.  Nothing in the reference calls them (its callers go to the factory directly), but the north star
names `service.embed_batch()` as a boundary, so they are kept, resolving the provider lazily on every call so that a
provider installed with `set_embedding_provider` (tests, explicit wiring) is honoured.
"""
from typing import List, Sequence

from rag.llm.embeddings import factory as _factory


def _provider():
    return _factory.get_embedding_provider()


async def embed(text: str) -> List[float]:
    """One text -> one vector (zero vector for empty / blank text, as every provider guarantees)."""
    vector = await _provider().embed_single(text)
    return vector


async def embed_batch(texts: Sequence[str]) -> List[List[float]]:
    """Texts -> vectors in the same order; an empty input returns an empty list."""
    items = list(texts)
    if not items:
        return []
    return await _provider().embed_batch(items)


def embedding_dim() -> int:
    """Width of the vectors the active provider produces."""
    return int(_provider().dimension())
