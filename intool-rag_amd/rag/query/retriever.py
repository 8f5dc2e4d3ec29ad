"""
rag/query/retriever.py -- the retriever the north star names; with hybrid search off it is the reference's
PageLevelRetriever (rag/query/page_retriever.py:78-288) step for step:

    embed query (no instruction, :109-110) -> vector search, limit = top_chunks = 50 (:117-121)
    -> RetrievedChunk list in search order (:123-139) -> group by page in first-seen order (:145-164)
    -> page score = mean(chunk scores) + min(0.05 * n, 0.15) in Python floats, stable sort descending (:166-213)
    -> first max_pages pages (:215-236)

With hybrid search on (HYBRID_SEARCH_ENABLED, rag/config.py:43) the chunk list is the reciprocal-rank fusion of the
dense list and a BM25 list over the same chunks (both depth top_chunks) -- the hybrid the reference's README.md:54-58
describes but never implements; spec in DESIGN.md.  Fusion order decides which chunks survive; each chunk keeps its
dense similarity as `score` so page scores stay on the reference's 0..1 scale; a chunk only the sparse leg found has
score 0.0 and metadata["sparse_only"], which keeps it out of its page's mean (it still counts for the chunk boost);
metadata["rrf_score"] / ["bm25_score"] carry the rest.

Dense search, BM25 and RRF run in libhiprag.so; grouping and page ranking are a few dozen Python float operations and
stay on the host exactly as the reference has them.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional

from rag.config import config
from rag.llm.embeddings.factory import get_embedding_provider
from rag.logging import logger


@dataclass
class RetrievedChunk:
    """Retrieved chunk with similarity score (page_retriever.py:25-32)."""
    chunk_id: str
    text: str
    score: float
    page: int
    metadata: Dict[str, Any]


@dataclass
class PageRanking:
    """Page with ranking score and its chunks (fields of page_retriever.py:35-41)."""
    page: int
    score: float
    chunks: List[RetrievedChunk]
    metadata: Dict[str, Any]

    # The two formatters the reference's response stage calls on every selected page (page_response.py:73 and :163);
    # their OUTPUT is pinned by tests/golden (page_formatting, produced by running the reference's own methods).
    _HEADING = (("chapter", "Chapter {}"), ("section", "Section {}"), ("title", "{}"))
    _CITED = (("chapter", "chapter"), ("section", "section"), ("subsection", "subsection"), ("title", "title"),
              ("source_file", "source_filename"))

    def get_context_text(self) -> str:
        """Prompt context of this page (same text as page_retriever.py:43-63): an optional `[Chapter c | Section s | title]`
        heading built from the truthy metadata fields, then the chunk texts, blank-line separated, outer whitespace stripped."""
        heading = " | ".join(fmt.format(self.metadata[key]) for key, fmt in self._HEADING if self.metadata.get(key))
        blocks = [f"[{heading}]"] if heading else []
        blocks.extend(chunk.text for chunk in self.chunks)
        return "\n\n".join(blocks).strip()

    def to_citation(self) -> Dict[str, Any]:
        """Citation record of this page (same keys and rounding as page_retriever.py:65-75)."""
        cite: Dict[str, Any] = {"page": self.page}
        for out_key, meta_key in self._CITED:
            cite[out_key] = self.metadata.get(meta_key)
        cite["relevance_score"] = round(self.score, 3)
        return cite


def group_chunks_by_page(chunks: List[RetrievedChunk]) -> Dict[int, List[RetrievedChunk]]:
    grouped: Dict[int, List[RetrievedChunk]] = {}
    for chunk in chunks:
        grouped.setdefault(chunk.page, []).append(chunk)
    return grouped


def rank_pages(chunks_by_page: Dict[int, List[RetrievedChunk]]) -> List[PageRanking]:
    """page score = mean(chunk scores) + min(0.05 n, 0.15), stable sort descending (page_retriever.py:166-213).
    Hybrid mode only: a chunk that ONLY the BM25 leg found has no dense similarity; it counts towards n (extra lexical
    evidence raises the page) but stays out of the mean -- averaging a 0.0 in would make a page rank LOWER for having
    more evidence.  With hybrid off no chunk is sparse-only and this is the reference's arithmetic unchanged."""
    rankings = []
    for page_num, page_chunks in chunks_by_page.items():
        scored = [c.score for c in page_chunks if not c.metadata.get("sparse_only")]
        avg_score = sum(scored) / len(scored) if scored else 0.0
        chunk_boost = min(len(page_chunks) * 0.05, 0.15)
        rankings.append(PageRanking(page=page_num, score=avg_score + chunk_boost, chunks=page_chunks,
                                    metadata=page_chunks[0].metadata))
    rankings.sort(key=lambda r: r.score, reverse=True)       # stable: ties keep first-seen page order
    return rankings


_META_KEYS = ("chapter", "section", "subsection", "title", "source_filename", "doc_id")    # page_retriever.py:129-136
_HYBRID_KEYS = ("rrf_score", "bm25_score", "sparse_only")


def _as_chunk(result: dict) -> RetrievedChunk:
    """One enriched search row -> RetrievedChunk with the reference's defaults (page_retriever.py:123-139)."""
    metadata: Dict[str, Any] = {key: result.get(key) for key in _META_KEYS}
    metadata.update((key, result[key]) for key in _HYBRID_KEYS if key in result)
    return RetrievedChunk(result.get("chunk_id", "unknown"), result.get("text", ""), result.get("score", 0),
                          result.get("page", 0), metadata)


class HybridRetriever:
    """Retrieve and rank at page level; `hybrid=None` follows HYBRID_SEARCH_ENABLED, False = the reference's path."""

    def __init__(self, top_chunks: int = 50, top_pages: int = 5, hybrid: Optional[bool] = False,
                 rrf_c: float = 60.0, weighted: bool = False):
        self.top_chunks = top_chunks
        self.top_pages = top_pages
        self.hybrid = config.HYBRID_SEARCH_ENABLED if hybrid is None else hybrid
        self.rrf_c = rrf_c
        # weighted RRF uses the reference's unused knobs VECTOR_WEIGHT / BM25_WEIGHT (config.py:44-45)
        self.w_dense, self.w_sparse = (config.VECTOR_WEIGHT, config.BM25_WEIGHT) if weighted else (1.0, 1.0)

    async def retrieve_chunks(self, query: str, project: Optional[str] = None) -> List[RetrievedChunk]:
        logger.info(f"Retrieving top-{self.top_chunks} chunks for query")
        embedding_provider = get_embedding_provider()
        query_embedding = await embedding_provider.embed_single(query)
        if self.hybrid:
            search_results = self._hybrid_search(query, query_embedding)
        else:
            from rag.storage.hip_index import search_hip_by_vector
            search_results = await search_hip_by_vector(query_embedding, limit=self.top_chunks, project=project)
        chunks = [_as_chunk(result) for result in search_results]
        logger.info(f"Retrieved {len(chunks)} chunks")
        return chunks

    def _hybrid_search(self, query: str, query_embedding: List[float]) -> List[dict]:
        """dense top-K + BM25 top-K over the same chunk rows -> RRF -> enriched dicts in fusion order."""
        import numpy as np
        from hiprag import rrf_fuse
        from rag.storage.hip_index import enrich, open_first_index
        from rag.storage.hip_index.sparse import get_sparse_index
        opened = open_first_index()
        if opened is None:
            logger.warning("No HIP indices found")
            return []
        reader, doc_id, chunks_list = opened
        depth = self.top_chunks
        dense = [(rid, sc) for rid, sc in reader.search(query_embedding, top_k=depth) if 0 <= rid < len(chunks_list)]
        dense_ids = np.full((1, depth), -1, dtype=np.int64)
        dense_ids[0, :len(dense)] = [rid for rid, _ in dense]
        dense_score = {rid: sc for rid, sc in dense}
        bm25 = get_sparse_index(config.STORAGE_DIR, doc_id, chunks_list)
        s_scores, s_ids = bm25.search([bm25.terms_of(query)], depth)
        f_scores, f_ids = rrf_fuse(dense_ids, s_ids, depth, c=self.rrf_c, w_a=self.w_dense, w_b=self.w_sparse)
        bm25_of = {int(i): float(s) for i, s in zip(s_ids[0], s_scores[0]) if i >= 0}
        fused = []
        for rid, fs in zip(f_ids[0], f_scores[0]):
            if rid < 0:
                continue
            rid = int(rid)
            item = enrich([(rid, dense_score.get(rid, 0.0))], chunks_list, compat_minus_one=False)[0]
            item["rrf_score"] = float(fs)
            if rid in bm25_of:
                item["bm25_score"] = bm25_of[rid]
            if rid not in dense_score:
                item["sparse_only"] = True       # no dense similarity: rank_pages keeps it out of the page mean
            fused.append(item)
        return fused

    # the three page-level steps keep the reference's method names
    def group_chunks_by_page(self, chunks: List[RetrievedChunk]) -> Dict[int, List[RetrievedChunk]]:
        return group_chunks_by_page(chunks)

    def rank_pages(self, chunks_by_page: Dict[int, List[RetrievedChunk]]) -> List[PageRanking]:
        return rank_pages(chunks_by_page)

    def select_top_pages(self, rankings: List[PageRanking], max_pages: Optional[int] = None) -> List[PageRanking]:
        max_pages = max_pages or self.top_pages
        selected = rankings[:max_pages]
        logger.info(f"Selected {len(selected)} pages from {len(rankings)} candidates")
        return selected

    async def retrieve_and_rank_pages(self, query: str, project: Optional[str] = None,
                                      max_pages: Optional[int] = None) -> List[PageRanking]:
        chunks = await self.retrieve_chunks(query, project)
        if not chunks:
            logger.warning("No chunks retrieved")
            return []
        return self.select_top_pages(self.rank_pages(self.group_chunks_by_page(chunks)), max_pages)


PageLevelRetriever = HybridRetriever      # the reference's class name (page_retriever.py:78)


async def retrieve_and_rank_pages(query: str, project: Optional[str] = None, top_pages: int = 5) -> List[PageRanking]:
    """Convenience function with the reference's signature (page_retriever.py:271-288)."""
    retriever = HybridRetriever(top_pages=top_pages)
    return await retriever.retrieve_and_rank_pages(query, project, top_pages)
