"""
Tokenisation for the hip provider.

Deployments point HIP_TOKENIZER_FILE at BGE-M3's `tokenizer.json` (HuggingFace `tokenizers` format; the wheel is
installed, the file is not shippable offline).  Without it `load_tokenizer` RAISES -- the reference raises when its model
cannot be loaded (rag/providers/hf/embeddings.py:26-29,39-40) and a service that answers with meaningless vectors is worse
than one that does not start.  HIP_ALLOW_SYNTHETIC=1 is the explicit opt-in (tests, benches) to a SYNTHETIC hashing
tokenizer -- lower-cased whitespace tokens (the reference's only tokenisation, rag/agent/query_processor.py:26) hashed
into the vocabulary -- which exercises the identical GPU path with meaningless ids.
Sequences are `<s> tokens </s>` (ids 0 / 2), truncated to max_seq_len like sentence-transformers does.
"""
from __future__ import annotations

import hashlib
import os
from typing import List, Optional

from rag.logging import logger


class HashTokenizer:
    synthetic = True

    def __init__(self, vocab: int, bos: int = 0, eos: int = 2, first_id: int = 3):
        self.vocab, self.bos, self.eos, self.first = vocab, bos, eos, first_id
        self._memo = {}
        logger.warning("[EMBED] SYNTHETIC hashing tokenizer in use: token ids carry no meaning")

    def _id(self, tok: str) -> int:
        hit = self._memo.get(tok)
        if hit is None:
            h = int.from_bytes(hashlib.blake2b(tok.encode("utf-8"), digest_size=8).digest(), "little")
            hit = self.first + h % (self.vocab - self.first)
            if len(self._memo) < 1_000_000:
                self._memo[tok] = hit
        return hit

    def encode(self, text: str, max_len: int) -> List[int]:
        body = [self._id(t) for t in text.lower().split()][:max(0, max_len - 2)]
        return [self.bos] + body + [self.eos]

    def encode_pair(self, a: str, b: str, max_len: int) -> List[int]:
        ta = [self._id(t) for t in a.lower().split()]
        tb = [self._id(t) for t in b.lower().split()]
        room = max(0, max_len - 4)
        ta = ta[:max(1, room // 4)] if len(ta) + len(tb) > room else ta     # queries are short; keep them whole if they fit
        tb = tb[:max(0, room - len(ta))]
        return [self.bos] + ta + [self.eos, self.eos] + tb + [self.eos]


class FileTokenizer:
    synthetic = False

    def __init__(self, path: str, bos: int = 0, eos: int = 2):
        from tokenizers import Tokenizer
        self.tk = Tokenizer.from_file(path)
        self.bos, self.eos = bos, eos

    def encode(self, text: str, max_len: int) -> List[int]:
        ids = self.tk.encode(text, add_special_tokens=False).ids[:max(0, max_len - 2)]
        return [self.bos] + ids + [self.eos]

    def encode_pair(self, a: str, b: str, max_len: int) -> List[int]:
        ia = self.tk.encode(a, add_special_tokens=False).ids
        ib = self.tk.encode(b, add_special_tokens=False).ids
        room = max(0, max_len - 4)
        ib = ib[:max(0, room - len(ia))]
        return [self.bos] + ia[:room] + [self.eos, self.eos] + ib + [self.eos]


def allow_synthetic() -> bool:
    """HIP_ALLOW_SYNTHETIC=1: random weights / the hashing tokenizer may stand in for missing model files."""
    return os.getenv("HIP_ALLOW_SYNTHETIC", "").lower() in ("1", "true", "yes")


def load_tokenizer(vocab: int, path: Optional[str] = None):
    path = path or os.getenv("HIP_TOKENIZER_FILE")
    if path:
        if not os.path.exists(path):
            raise RuntimeError(f"HIP_TOKENIZER_FILE={path!r} does not exist")
        return FileTokenizer(path)
    if not allow_synthetic():
        raise RuntimeError("HIP_TOKENIZER_FILE is not set: the hip provider needs the model's tokenizer.json "
                           "(set HIP_ALLOW_SYNTHETIC=1 to run on the synthetic hashing tokenizer)")
    return HashTokenizer(vocab)
