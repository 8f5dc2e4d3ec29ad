#!/usr/bin/env python3
"""
Generates tests/golden/reference_wrapper_golden.json by RUNNING THE REFERENCE'S OWN PYTHON
(/root/reference on PYTHONPATH) for the parts of the hot path whose code exists there:

  1. FAISSIndexReader.search          rag/storage/faiss_index.py:63-91   (score transform, clamp, -1 passthrough)
  2. search_faiss_by_vector            rag/storage/faiss_index.py:137-199 (first-index pick, chunk enrichment,
                                                                           the -1 -> LAST chunk quirk)
  3. group/rank/select pages           rag/query/page_retriever.py:145-236
  3b. PageRanking text / citation      rag/query/page_retriever.py:43-75
  4. FileStorageManager JSON schemas   rag/storage/file_storage.py:87-166,194-252

Run (in the build container only; /root/reference does not exist on the GPU box):
    python tests/golden/make_reference_golden.py

What is and is not the reference here
-------------------------------------
faiss-cpu is not installed, so the FAISS *arithmetic* cannot run; `index.search` is replaced by an object that
returns (D, I) arrays prepared by this script (hand-written distances, or distances from oracle/'s fp64 truth
rounded to fp32).  Everything downstream of those arrays -- the code being pinned -- is the reference's own.
The dense arithmetic itself stays PARITY UNPINNED.

numpy note: the reference pins numpy<2.0.0 (rag/requirements.txt:15) where `np.float32 - python float`
promotes to float64; this image has numpy 2.2 (NEP 50 keeps float32).  To reproduce the pinned environment's
arithmetic the stub hands the distances over as a float64 array holding float32-representable values.
"""
import asyncio
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

tmp = tempfile.mkdtemp(prefix="refgolden_")
os.environ["STORAGE_DIR"] = os.path.join(tmp, "storages")
os.environ["CACHE_DIR"] = os.path.join(tmp, "cache")
os.chdir(tmp)
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import rag.storage.faiss_index as fi            # noqa: E402  (reference)
import rag.query.page_retriever as pr           # noqa: E402  (reference)
from rag.storage.file_storage import FileStorageManager  # noqa: E402  (reference)
from rag.ingest.schemas import Chunk            # noqa: E402  (reference)
from rag.config import config                   # noqa: E402  (reference)

from oracle import hybrid_oracle as ho          # noqa: E402  (supplies D, I only)

FLT_MAX = float(np.finfo(np.float32).max)


class PreparedIndex:
    """Stands where a faiss.Index would: returns the (D, I) it was given.  Not a FAISS implementation."""

    def __init__(self, d, ntotal, D=None, I=None, vectors=None):
        self.d, self.ntotal, self._D, self._I, self._x = d, ntotal, D, I, vectors

    def search(self, q, k):
        if self._x is not None:     # distances from the oracle's fp64 truth, rounded to fp32 like FAISS returns
            s, i = ho.flat_search(self._x, np.asarray(q, np.float32), k, ho.METRIC_L2)
            return s.astype(np.float64), i
        D = np.asarray(self._D, dtype=np.float32).astype(np.float64)[None, :k]
        I = np.asarray(self._I, dtype=np.int64)[None, :k]
        return D, I


def reader_with(index):
    r = object.__new__(fi.FAISSIndexReader)
    r.index_path, r.index = "<prepared>", index
    return r


golden = {"generator": "tests/golden/make_reference_golden.py", "reference": "batd-htplus/intool-rag",
          "numpy": np.__version__}

# ---- 1. FAISSIndexReader.search ----------------------------------------------------------------
cases = []
prepared = [
    ([0.0, 0.25, 1.0, 1.6442, 1.8014, 1.8554, 2.0, 2.5, 3.999], [2, 1, 0, 4, 3, 7, 9, 8, 5]),
    ([0.5, FLT_MAX, FLT_MAX], [0, -1, -1]),                      # k > ntotal padding
    ([-0.25, 1e-7, 1.9999999], [3, 1, 2]),                       # negative "distance" clamps to 1
    ([0.7853982, 1.5707964, 0.33333334, 1.2345679], [11, 5, 6, 1]),
    ([], []),
]
for D, I in prepared:
    k = len(D)
    res = reader_with(PreparedIndex(4, 10, D, I)).search([0.0, 0.0, 0.0, 0.0], top_k=k) if k else []
    cases.append({"distances_f32": [float(np.float32(v)) for v in D], "ids": I, "expected": [[a, b] for a, b in res]})
golden["reader_search"] = cases

# ---- 2. search_faiss_by_vector ------------------------------------------------------------------
rng = np.random.default_rng(20240101)
n, d = 23, 16
vecs = rng.standard_normal((n, d)).astype(np.float32)
vecs /= np.linalg.norm(vecs, axis=1, keepdims=True)
doc_id = "docA"
chunks = [Chunk(chunk_id=f"c_{1 + i // 4:03d}_{i % 4:03d}", node_id=f"{i:04d}", page=1 + i // 4,
                text=f"chunk text number {i}", seq_index=i % 4) for i in range(n)]
storage = FileStorageManager(str(config.STORAGE_DIR))
chunks_path = storage.save_chunks(chunks, doc_id=doc_id)
meta_path = storage.save_faiss_metadata(chunks, doc_id=doc_id)
open(os.path.join(str(config.STORAGE_DIR), f"{doc_id}_faiss.index"), "wb").write(b"prepared")
with open(chunks_path) as f:
    chunks_json = json.load(f)
with open(meta_path) as f:
    meta_json = json.load(f)
loaded = storage.load_chunks(doc_id)

# Route FAISSIndexReader's loader to the prepared index (the reference module has HAS_FAISS=False here).
fi.HAS_FAISS = True
fi.faiss = types.SimpleNamespace(read_index=lambda path: PreparedIndex(d, n, vectors=vecs))
fi._INDEX_CACHE.clear()

sfv = []
queries = [vecs[3] + 0.05 * rng.standard_normal(d).astype(np.float32) for _ in range(3)]
queries = [q / np.linalg.norm(q) for q in queries]
for q, limit in zip(queries, (5, 23, 30)):      # limit 30 > ntotal exercises the -1 -> last-chunk quirk
    out = asyncio.run(fi.search_faiss_by_vector([float(v) for v in q], limit=limit))
    sfv.append({"query": [float(v) for v in q], "limit": limit, "expected": out})
golden["search_faiss_by_vector"] = {
    "vectors": [[float(v) for v in row] for row in vecs], "doc_id": doc_id,
    "chunks_json": chunks_json, "faiss_meta_json": meta_json,
    "loaded_chunk_ids_in_order": list(loaded.keys()), "cases": sfv,
}
fi._INDEX_CACHE.clear()
# no-index case (faiss_index.py:163-165)
for fn in os.listdir(str(config.STORAGE_DIR)):
    if fn.endswith("_faiss.index"):
        os.remove(os.path.join(str(config.STORAGE_DIR), fn))
golden["search_faiss_by_vector"]["no_index_expected"] = asyncio.run(fi.search_faiss_by_vector([0.0] * d, limit=5))

# ---- 3. page grouping / ranking -----------------------------------------------------------------
R = pr.RetrievedChunk


def run_pages(spec, max_pages):
    chunks_ = [R(chunk_id=f"c{i}", text=f"t{i}", score=s, page=p, metadata={"title": f"T{p}"}) for i, (s, p) in enumerate(spec)]
    r = pr.PageLevelRetriever(top_pages=max_pages)
    ranked = r.select_top_pages(r.rank_pages(r.group_chunks_by_page(chunks_)), max_pages)
    return [{"page": x.page, "score": x.score, "chunk_ids": [c.chunk_id for c in x.chunks],
             "citation_score": x.to_citation()["relevance_score"]} for x in ranked]


page_cases = []
specs = [
    ([(0.9, 1), (0.5, 2), (0.7, 1), (0.9, 3), (0.0, 2), (0.0, 2), (0.0, 2)], 5),
    ([(0.6, 4), (0.6, 2), (0.6, 9)], 5),                                    # exact ties keep first-seen order
    ([(0.61, 1), (0.59, 1), (0.60, 1), (0.62, 1), (0.58, 1), (0.8, 2)], 1),  # boost cap 0.15 at n>=3
    ([(1.0, 7)], 3),
]
rs = np.random.default_rng(7)
rand_spec = [(float(np.float32(rs.random())), int(rs.integers(1, 12))) for _ in range(50)]
specs.append((rand_spec, 5))
for spec, mp in specs:
    page_cases.append({"chunks": [[s, p] for s, p in spec], "max_pages": mp, "expected": run_pages(spec, mp)})
golden["page_ranking"] = page_cases

# ---- 3b. PageRanking.get_context_text / to_citation (page_retriever.py:43-75; consumed by page_response.py:73,163) ----
fmt_cases = []
fmt_specs = [
    ({"chapter": "3", "section": "3.2", "subsection": "3.2.1", "title": "Cooling loop", "source_filename": "manual.pdf"},
     ["first chunk", "second chunk with trailing space ", ""], 4, 0.81249),
    ({"title": "Only a title"}, ["alone"], 1, 1.15),
    ({"chapter": None, "section": "", "title": None, "source_filename": None}, ["  padded text\n", "x"], 9, 0.0004),
    ({"chapter": 7, "section": 0, "title": "Zero section is falsy"}, [], 2, 0.5555),
    ({}, ["no metadata at all", "two\n\nparagraphs"], 12, 0.33333333),
]
for meta, texts, page, score in fmt_specs:
    chunks_ = [R(chunk_id=f"c{i}", text=t, score=score, page=page, metadata=meta) for i, t in enumerate(texts)]
    pg = pr.PageRanking(page=page, score=score, chunks=chunks_, metadata=meta)
    fmt_cases.append({"metadata": meta, "texts": texts, "page": page, "score": score,
                      "context_text": pg.get_context_text(), "citation": pg.to_citation()})
golden["page_formatting"] = fmt_cases

out_path = os.path.join(HERE, "reference_wrapper_golden.json")
with open(out_path, "w") as f:
    json.dump(golden, f, indent=1)
print("wrote", out_path)
