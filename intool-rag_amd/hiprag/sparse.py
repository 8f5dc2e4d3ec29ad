"""
HipBM25 -- BM25 term-at-a-time scoring on the GPU (csrc/bm25.hip) plus the host-side index builder.

The reference only names BM25 (README.md:54-58, rag/config.py:43-45); the specification implemented here is in
DESIGN.md ("BM25 spec"): tokens = text.lower().split() (the reference's only tokeniser,
rag/agent/query_processor.py:26); idf = ln(1 + (N - df + 0.5)/(df + 0.5)); impact = idf * tf*(k1+1) /
(tf + k1*(1 - b + b*|d|/avgdl)) with k1=1.5, b=0.75, evaluated in float64 and rounded ONCE to float32.  The GPU
adds impacts in query-term order (fp32), excludes score <= 0 and ranks by (score desc, doc id asc).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _native as nat

K1 = 1.5
B = 0.75


def tokenize(text: str) -> List[str]:
    return text.lower().split()


@dataclass
class PostingsCSR:
    n_docs: int
    n_terms: int
    offsets: np.ndarray   # uint64 [V+1]
    doc_ids: np.ndarray   # uint32 [P], ascending inside each term's list
    impacts: np.ndarray   # float32 [P]
    vocab: Optional[Dict[str, int]] = None

    def shard(self, lo: int, hi: int) -> "PostingsCSR":
        """Postings of documents [lo,hi) with LOCAL doc ids; impacts keep the GLOBAL idf / avgdl."""
        keep = (self.doc_ids >= lo) & (self.doc_ids < hi)
        term_of = np.repeat(np.arange(self.n_terms, dtype=np.int64), np.diff(self.offsets.astype(np.int64)))
        df = np.bincount(term_of[keep], minlength=self.n_terms)
        off = np.zeros(self.n_terms + 1, dtype=np.uint64)
        off[1:] = np.cumsum(df).astype(np.uint64)
        return PostingsCSR(hi - lo, self.n_terms, off, (self.doc_ids[keep] - np.uint32(lo)).astype(np.uint32),
                           self.impacts[keep].copy(), self.vocab)


def build_postings(doc_of_tok: np.ndarray, term_of_tok: np.ndarray, n_docs: int, n_terms: int,
                   doc_len: Optional[np.ndarray] = None, k1: float = K1, b: float = B, *,
                   df_global: Optional[np.ndarray] = None, n_docs_global: Optional[int] = None,
                   avgdl_global: Optional[float] = None) -> PostingsCSR:
    """Token stream -> CSR postings (tf = multiplicity) with precomputed fp32 impacts.

    A document-range SHARD built on its own (one rank of a row-sharded collection) passes the collection-wide document
    frequencies, document count and average length (`df_global`, `n_docs_global`, `avgdl_global`): idf and the length
    normalisation are global constants of BM25, so the shard's impacts then equal those of the unsharded build."""
    doc_of_tok = np.asarray(doc_of_tok, dtype=np.int64)
    term_of_tok = np.asarray(term_of_tok, dtype=np.int64)
    if doc_len is None:
        doc_len = np.bincount(doc_of_tok, minlength=n_docs)
    doc_len = np.asarray(doc_len, dtype=np.float64)
    pair = np.sort(term_of_tok * np.int64(n_docs) + doc_of_tok, kind="stable")
    if pair.size:
        first = np.ones(pair.size, dtype=bool)
        first[1:] = pair[1:] != pair[:-1]
        starts = np.flatnonzero(first)
        uniq = pair[starts]
        tf = np.diff(np.append(starts, pair.size)).astype(np.float64)
    else:
        uniq = pair
        tf = np.zeros(0, dtype=np.float64)
    term = uniq // n_docs if n_docs else uniq
    doc = (uniq - term * n_docs).astype(np.uint32)
    df = np.bincount(term, minlength=n_terms).astype(np.float64)
    offsets = np.zeros(n_terms + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(df).astype(np.uint64)
    avgdl = float(avgdl_global) if avgdl_global is not None else (doc_len.sum() / n_docs if n_docs else 1.0)
    n_all = float(n_docs_global) if n_docs_global is not None else float(n_docs)
    df_all = np.asarray(df_global, dtype=np.float64) if df_global is not None else df
    idf = np.log(1.0 + (n_all - df_all + 0.5) / (df_all + 0.5))
    norm = k1 * (1.0 - b + b * doc_len[doc] / avgdl)
    impacts = (idf[term] * tf * (k1 + 1.0) / (tf + norm)).astype(np.float32)
    return PostingsCSR(n_docs, n_terms, offsets, doc, impacts)


def build_postings_from_texts(texts: Sequence[str]) -> PostingsCSR:
    vocab: Dict[str, int] = {}
    docs: List[int] = []
    terms: List[int] = []
    for i, t in enumerate(texts):
        for tok in tokenize(t or ""):
            terms.append(vocab.setdefault(tok, len(vocab)))
            docs.append(i)
    n = len(texts)
    p = build_postings(np.asarray(docs, np.int64), np.asarray(terms, np.int64), n, max(len(vocab), 1),
                       np.bincount(np.asarray(docs, np.int64), minlength=n) if docs else np.zeros(n))
    p.vocab = vocab
    return p


class HipBM25:
    def __init__(self, postings: PostingsCSR, device: int = 0, id_base: int = 0):
        self.p = postings
        self.device = int(device)
        off = np.ascontiguousarray(postings.offsets, dtype=np.uint64)
        ids = np.ascontiguousarray(postings.doc_ids, dtype=np.uint32)
        imp = np.ascontiguousarray(postings.impacts, dtype=np.float32)
        if off.shape[0] != postings.n_terms + 1 or ids.shape != imp.shape or int(off[-1]) != ids.shape[0]:
            raise ValueError("inconsistent CSR postings")
        h = ctypes.c_uint64()
        nat.call("hipbm25_create", int(postings.n_docs), int(postings.n_terms), off.ctypes.data, ids.ctypes.data,
                 imp.ctypes.data, self.device, ctypes.byref(h))
        self._h = h.value
        if id_base:
            nat.call("hipbm25_set_id_base", self._h, int(id_base))

    def set_id_base(self, base: int) -> None:
        """First global document id of this (document-range) shard: returned ids are local id + base."""
        nat.call("hipbm25_set_id_base", self._h, int(base))

    def close(self) -> None:
        if getattr(self, "_h", None):
            try:
                nat.call("hipbm25_destroy", self._h)
            finally:
                self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _flatten(queries: Sequence[Sequence[int]]) -> Tuple[np.ndarray, np.ndarray]:
        qoff = np.zeros(len(queries) + 1, dtype=np.int32)
        qoff[1:] = np.cumsum([len(q) for q in queries])
        terms = (np.concatenate([np.asarray(q, dtype=np.uint32) for q in queries]) if qoff[-1]
                 else np.zeros(0, np.uint32))
        return np.ascontiguousarray(terms, dtype=np.uint32), qoff

    def terms_of(self, text: str) -> List[int]:
        """Query text -> term ids in query order; out-of-vocabulary tokens are dropped (they match nothing)."""
        if self.p.vocab is None:
            raise ValueError("index was built from term ids, not texts")
        return [self.p.vocab[t] for t in tokenize(text) if t in self.p.vocab]

    def search(self, queries: Sequence[Sequence[int]], k: int) -> Tuple[np.ndarray, np.ndarray]:
        terms, qoff = self._flatten(queries)
        nq = len(queries)
        scores = np.empty((nq, k), dtype=np.float32)
        ids = np.empty((nq, k), dtype=np.int64)
        nat.call("hipbm25_search", self._h, terms.ctypes.data if terms.size else None, qoff.ctypes.data, nq, int(k),
                 scores.ctypes.data, ids.ctypes.data)
        return scores, ids

    def search_device(self, queries: Sequence[Sequence[int]], k: int, out=None):
        import torch
        from .index import _stream_ptr
        terms, qoff = self._flatten(queries)
        nq = len(queries)
        if out is None:
            dev = torch.device("cuda", self.device)
            out = (torch.empty((nq, k), dtype=torch.float64, device=dev),
                   torch.empty((nq, k), dtype=torch.float32, device=dev),
                   torch.empty((nq, k), dtype=torch.int64, device=dev))
        nat.call("hipbm25_search_dev", self._h, terms.ctypes.data if terms.size else None, qoff.ctypes.data, nq, int(k),
                 out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), _stream_ptr())
        return out

    def stats(self) -> dict:
        st = nat.HipBm25Stats()
        nat.call("hipbm25_get_stats", self._h, ctypes.byref(st))
        return {f: getattr(st, f) for f, _ in st._fields_}
