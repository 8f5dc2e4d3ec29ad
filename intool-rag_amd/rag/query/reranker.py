"""
rag/query/reranker.py -- cross-encoder re-ranking on the GPU encoder.

The reference configures a reranker (RERANKER_MODEL = BAAI/bge-reranker-v2-m3, RERANKER_ENABLED, RERANKER_TOP_K = 10,
rag/config.py:25-27) and declares RerankerError (rag/core/exceptions.py:25) but contains no reranking code; this module
is the build-defined implementation: each (query, chunk) pair becomes `<s> query </s></s> passage </s>`
(truncated to 512 tokens), the XLM-R encoder + classification head produce one logit per pair
(csrc/encoder.hip rerank_head_kernel), and chunks are returned by (logit descending, original retrieval order).
"""
from __future__ import annotations

import asyncio
import os
import threading
from typing import List, Optional

from rag.config import config
from rag.logging import logger
from rag.providers.hip.tokenizer import allow_synthetic, load_tokenizer


class RerankerError(Exception):
    """Name taken from the reference's unused rag/core/exceptions.py:25."""


class CrossEncoderReranker:
    def __init__(self, encoder=None, tokenizer=None, top_k: Optional[int] = None):
        try:
            from hiprag import EncoderConfig, HipEncoder
        except Exception as e:
            raise RerankerError(f"libhiprag not available: {e}")
        if encoder is None:
            cfg = EncoderConfig()
            weights = os.getenv("HIP_RERANKER_WEIGHTS")
            state = None
            if weights:
                from safetensors.torch import load_file
                state = load_file(weights)
            elif not allow_synthetic():
                raise RerankerError("HIP_RERANKER_WEIGHTS is not set: the reranker needs a local safetensors file of "
                                    f"{config.RERANKER_MODEL} (set HIP_ALLOW_SYNTHETIC=1 to run on seeded random weights)")
            else:
                logger.warning("[RERANK] HIP_ALLOW_SYNTHETIC: seeded RANDOM weights of the "
                               f"{config.RERANKER_MODEL} architecture -- logits carry no meaning")
            if tokenizer is None:
                try:
                    tokenizer = load_tokenizer(cfg.vocab)
                except RuntimeError as e:
                    raise RerankerError(str(e))
            encoder = HipEncoder(cfg, state, device=config.HIP_DEVICE, seed=1, with_head=True)
        if not encoder.has_head:
            raise RerankerError("the reranker needs an encoder with a classification head")
        self.encoder = encoder
        self.tokenizer = tokenizer or load_tokenizer(encoder.cfg.vocab)
        self.top_k = top_k or config.RERANKER_TOP_K
        self._lock = threading.Lock()

    def score(self, query: str, passages: List[str]) -> List[float]:
        pairs = [self.tokenizer.encode_pair(query.replace("\n", " "), p.replace("\n", " "), self.encoder.cfg.max_seq_len)
                 for p in passages]
        with self._lock:
            return self.encoder.score_tokens(pairs, batch_size=4096, max_tokens=256 * 512).cpu().tolist()

    async def rerank(self, query: str, chunks: list, top_k: Optional[int] = None) -> list:
        """chunks: RetrievedChunk list in retrieval order -> the top_k by cross-encoder logit; each returned chunk carries
        metadata["rerank_score"]."""
        if not chunks:
            return []
        try:
            scores = await asyncio.to_thread(self.score, query, [c.text for c in chunks])
        except Exception as e:
            logger.error(f"[RERANK] failed: {e}")
            raise RerankerError(str(e))
        order = sorted(range(len(chunks)), key=lambda i: (-scores[i], i))[:top_k or self.top_k]
        out = []
        for i in order:
            chunks[i].metadata["rerank_score"] = float(scores[i])
            out.append(chunks[i])
        return out
