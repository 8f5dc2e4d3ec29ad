"""
rag/providers/hip/embeddings.py -- MI355X embedding provider behind the reference's EmbeddingProvider ABC
(rag/llm/embeddings/base.py:5-17), selected with EMBEDDING_PROVIDER=hip.

Behaviour mirrors HuggingFaceEmbeddingProvider (rag/providers/hf/embeddings.py:13-91) call for call:
  * embed_single: strip; empty / blank -> [0.0] * dim (:47-48); else encode in a worker thread (asyncio.to_thread, :53)
  * embed_batch : [] -> []; per-text strip, None / blank -> "" which IS encoded (:70-73); one batched encode (:76);
                  `instruction` and `batch_size` accepted and ignored exactly like the reference (:45, :64-65);
                  HIP_APPLY_INSTRUCTION=true prepends a given instruction instead (rag/config.py:53-60 defines them)
  * LangChain's HuggingFaceEmbeddings replaces "\\n" by " " before encoding and asks sentence-transformers for
    normalize_embeddings=True (:34): CLS pooling + L2 normalisation happen on the GPU (csrc/encoder.hip pool_kernel)
  * dimension(): the model width (the reference probes it with a dummy encode, :37-38)

Weights: HIP_ENCODER_WEIGHTS = path to a local safetensors file with XLM-R parameter names (BAAI/bge-m3 layout) and
HIP_TOKENIZER_FILE = the model's tokenizer.json.  Without them the constructor RAISES, like the reference does when its
model cannot be loaded (:26-29, :39-40).  HIP_ALLOW_SYNTHETIC=1 opts in to SEEDED RANDOM weights of the configured
architecture and the hashing tokenizer (tests and benches: there is no checkpoint in an offline image; the GPU code
path, shapes and cost are identical, the vectors are meaningless).
"""
from __future__ import annotations

import asyncio
import json
import os
import threading
from typing import List, Optional

from rag.config import config
from rag.llm.embeddings.base import EmbeddingProvider
from rag.logging import logger
from rag.providers.hip.tokenizer import allow_synthetic, load_tokenizer

try:
    from hiprag import EncoderConfig, HipEncoder
    HAS_HIP = True
    _ERR: Optional[Exception] = None
except Exception as _e:
    HAS_HIP = False
    _ERR = _e


# One forward takes up to this many padded tokens (the BASELINE config-5 batch, 256 x 512): batches are cut by token count,
# not by sequence count, so that short chunks fill the GPU as well as long ones (sentence-transformers' 32 sequences of
# ~128 tokens are a twentieth of it).  EMBEDDING_BATCH_SIZE is accepted and ignored like in the reference (:64-65).
_MAX_BATCH_TOKENS = int(os.getenv("HIP_ENCODER_BATCH_TOKENS", str(256 * 512)))
_MAX_BATCH_SEQS = 4096


def _encoder_config() -> "EncoderConfig":
    cfg = EncoderConfig()                                  # XLM-R large = BAAI/bge-m3 (config.EMBEDDING_MODEL)
    override = os.getenv("HIP_ENCODER_CONFIG")             # JSON, e.g. {"layers": 2, "hidden": 256, ...} for tests
    if override:
        for k, v in json.loads(override).items():
            setattr(cfg, k, v)
    return cfg


class HipEmbeddingProvider(EmbeddingProvider):
    def __init__(self, model_name: Optional[str] = None, encoder: Optional["HipEncoder"] = None, tokenizer=None):
        if not HAS_HIP:
            raise RuntimeError(f"libhiprag not available: {_ERR}")
        self.model_name = model_name or config.EMBEDDING_MODEL
        if encoder is None:
            cfg = _encoder_config()
            weights = os.getenv("HIP_ENCODER_WEIGHTS")
            state = None
            if weights:
                from safetensors.torch import load_file
                state = load_file(weights)
            elif not allow_synthetic():
                raise RuntimeError("HIP_ENCODER_WEIGHTS is not set: the hip embedding provider needs a local safetensors "
                                   f"file of {self.model_name} (set HIP_ALLOW_SYNTHETIC=1 to run on seeded random weights)")
            else:
                logger.warning("[EMBED] HIP_ALLOW_SYNTHETIC: seeded RANDOM weights of the "
                               f"{self.model_name} architecture -- embeddings carry no meaning")
            if tokenizer is None:
                tokenizer = load_tokenizer(cfg.vocab)      # raises before any GPU memory is taken
            encoder = HipEncoder(cfg, state, device=config.HIP_DEVICE, seed=0)
        self.encoder = encoder
        self.tokenizer = tokenizer or load_tokenizer(encoder.cfg.vocab)
        self._dimension = encoder.dimension
        self._lock = threading.Lock()      # to_thread workers share one encoder workspace
        logger.info(f"[EMBED] ✓ HIP ready (model={self.model_name}, dim={self._dimension})")

    def _encode(self, texts: List[str]) -> List[List[float]]:
        toks = [self.tokenizer.encode(t.replace("\n", " "), self.encoder.cfg.max_seq_len) for t in texts]
        with self._lock:
            out = self.encoder.encode_tokens(toks, batch_size=_MAX_BATCH_SEQS, max_tokens=_MAX_BATCH_TOKENS)
            return out.cpu().tolist()

    async def embed_single(self, text: str, instruction: Optional[str] = None) -> List[float]:
        if not text or not text.strip():
            return [0.0] * self._dimension
        clean_text = text.strip()
        if instruction and config.HIP_APPLY_INSTRUCTION:
            clean_text = instruction + clean_text
        try:
            vectors = await asyncio.to_thread(self._encode, [clean_text])
            return vectors[0]
        except Exception as e:
            logger.error(f"[EMBED] HIP embed_single failed: {e}")
            raise

    async def embed_batch(self, texts: List[str], instruction: Optional[str] = None, batch_size: int = 32) -> List[List[float]]:
        if not texts:
            return []
        clean_texts = [t.strip() if t and t.strip() else "" for t in texts]
        if instruction and config.HIP_APPLY_INSTRUCTION:
            clean_texts = [instruction + t if t else t for t in clean_texts]
        try:
            vectors = await asyncio.to_thread(self._encode, clean_texts)
            return [v if v else [0.0] * self._dimension for v in vectors]
        except Exception as e:
            logger.error(f"[EMBED] HIP embed_batch failed: {e}")
            raise

    async def embed_batch_device(self, texts: List[str]):
        """embed_batch without the list-of-lists round trip: a float32 CUDA tensor [n, dim] (the ingest side feeds it
        straight into the index, rag/ingest/indexing.py).  Same text cleaning as embed_batch."""
        import torch
        if not texts:
            return torch.empty((0, self._dimension), dtype=torch.float32, device=f"cuda:{self.encoder.device}")
        clean_texts = [t.strip() if t and t.strip() else "" for t in texts]

        def run():
            toks = [self.tokenizer.encode(t.replace("\n", " "), self.encoder.cfg.max_seq_len) for t in clean_texts]
            with self._lock:
                return self.encoder.encode_tokens(toks, batch_size=_MAX_BATCH_SEQS, max_tokens=_MAX_BATCH_TOKENS)
        return await asyncio.to_thread(run)

    def dimension(self) -> int:
        return self._dimension
