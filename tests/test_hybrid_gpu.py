"""
GPU parity of the sparse leg (BM25 TAAT), rank fusion (RRF) and the rag/ overlay's search functions against the CPU
oracle and the reference's golden outputs.  Bar: ids bit-exact; BM25 / RRF scores bit-exact fp32 (same op order).
"""
import asyncio
import json
import os

import numpy as np
import pytest

from oracle import hybrid_oracle as ho

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_wrapper_golden.json")))


def _gpu_postings(p):
    from hiprag import PostingsCSR
    return PostingsCSR(p.n_docs, p.n_terms, p.offsets, p.doc_ids, p.impacts)


@pytest.mark.parametrize("n_docs,n_terms,k", [(20000, 4096, 50), (5000, 512, 10), (300, 64, 7)])
def test_bm25_taat_matches_oracle_bit_exact(gpu, n_docs, n_terms, k):
    from hiprag import HipBM25
    p = ho.synthetic_postings(n_docs, n_terms=n_terms, seed=777)
    queries = ho.synthetic_sparse_queries(37, n_terms=n_terms, terms_per_query=6, seed=888, min_rank=min(16, n_terms // 4))
    queries[3] = np.asarray([queries[3][0], queries[3][0], queries[3][1]], dtype=np.uint32)   # duplicate term adds twice
    queries[4] = np.asarray([], dtype=np.uint32)                                             # empty query
    queries[5] = np.asarray([n_terms + 5, queries[5][0]], dtype=np.uint32)                   # unknown term is skipped
    es, ei = ho.bm25_search(p, queries, k)
    ix = HipBM25(_gpu_postings(p))
    s, i = ix.search(queries, k)
    assert np.array_equal(i, ei)
    assert np.array_equal(s, es)
    st = ix.stats()
    assert st["queries"] == len(queries) and st["postings_touched"] > 0


def test_bm25_ties_and_zero_scores(gpu):
    from hiprag import HipBM25, build_postings_from_texts
    texts = ["alpha beta"] * 40 + ["gamma"] * 5 + ["alpha"] * 3        # many exact score ties
    p = build_postings_from_texts(texts)
    ix = HipBM25(p)
    op = ho.build_postings_from_texts(texts)
    q = [ix.terms_of("alpha"), ix.terms_of("gamma delta"), ix.terms_of("nothing here")]
    assert q[2] == []
    s, i = ix.search(q, 10)
    es, ei = ho.bm25_search(op, q, 10)
    assert np.array_equal(i, ei) and np.array_equal(s, es)
    assert i[1].tolist()[:5] == [40, 41, 42, 43, 44] and i[1, 5] == -1     # only 5 docs score > 0
    assert np.all(i[2] == -1)


def test_bm25_mass_ties_across_tiles(gpu):
    """A few distinct texts repeated over two document tiles: thousands of exact score ties per wave (more survivors than
    the compaction buffer holds -> the sorted-list path) next to queries with a handful of survivors (the compaction +
    rank path); ids and scores equal to the oracle's at k below, at and above the per-wave list sizes."""
    from hiprag import HipBM25, build_postings_from_texts
    rng = np.random.default_rng(31)
    words = [f"w{j}" for j in range(12)]
    base = [" ".join(rng.choice(words, size=int(rng.integers(1, 5)))) for _ in range(9)]
    texts = [base[int(t)] for t in rng.integers(0, len(base), size=20000)]
    texts[777] = "w0 w1 w2 rareterm"
    texts[19000] = "rareterm rareterm w3"
    p = build_postings_from_texts(texts)
    op = ho.build_postings_from_texts(texts)
    ix = HipBM25(p)
    q = [ix.terms_of(" ".join(rng.choice(words, size=3))) for _ in range(5)] + [ix.terms_of("rareterm"), ix.terms_of("rareterm w3")]
    for k in (5, 50, 64):
        s, i = ix.search(q, k)
        es, ei = ho.bm25_search(op, q, k)
        assert np.array_equal(i, ei) and np.array_equal(s, es), k


def test_bm25_doc_range_shards_merge_to_unsharded(gpu):
    import torch
    from hiprag import HipBM25, merge_topk_device
    p = ho.synthetic_postings(9000, n_terms=1024, seed=5)
    queries = ho.synthetic_sparse_queries(9, n_terms=1024, seed=6)
    k = 20
    es, ei = ho.bm25_search(p, queries, k)
    full = _gpu_postings(p)
    bounds = [(0, 2500), (2500, 2501), (2501, 9000)]
    ps, pi = [], []
    shards = []
    for lo, hi in bounds:
        ix = HipBM25(full.shard(lo, hi), id_base=lo)
        s64, s32, ids = ix.search_device(queries, k)
        ps.append(s64)
        pi.append(ids)
        shards.append(ix)
    m64, m32, mids = merge_topk_device(torch.stack(ps), torch.stack(pi), k, "ip")
    torch.cuda.synchronize()
    assert np.array_equal(mids.cpu().numpy(), ei)
    assert np.array_equal(m32.cpu().numpy(), es)


def test_row_sharded_hybrid_equals_unsharded(gpu):
    """SURVEY 8e / configs[3]: dense rows and postings split by the same document ranges, both partial lists in one
    packed buffer (what the single all-gather carries), each leg merged globally, RRF after the merge -- identical to the
    unsharded hybrid result; and ShardedHybrid itself at world size 1."""
    import torch
    from hiprag import HipBM25, HipFlatIndex
    from hiprag.sharded import EmulatedHybridShards, ShardedHybrid
    n, d, depth, k = 9000, 64, 50, 10
    x = ho.synthetic_vectors(n, d, seed=51)
    q = ho.synthetic_queries(7, d, seed=52)
    p = ho.synthetic_postings(n, n_terms=512, seed=53)
    sq = ho.synthetic_sparse_queries(7, n_terms=512, terms_per_query=5, seed=54, min_rank=4)
    full = _gpu_postings(p)
    _, di = ho.flat_search(x, q, depth, ho.METRIC_L2)
    _, bi = ho.bm25_search(p, sq, depth)
    qd = torch.from_numpy(q).cuda()
    for (c, wd, ws) in [(60.0, 1.0, 1.0), (60.0, 0.7, 0.3)]:
        es, ei = ho.rrf_fuse(di, bi, k, c=c, w_a=wd, w_b=ws)
        bounds = [(0, 2500), (2500, 2501), (2501, 6000), (6000, 9000)]
        shards = []
        for lo, hi in bounds:
            ix = HipFlatIndex(d, "l2")
            ix.add(x[lo:hi])
            shards.append((ix, HipBM25(full.shard(lo, hi))))
        fs, fi = EmulatedHybridShards(shards, bounds).search_device(qd, sq, depth=depth, k=k, c=c, w_dense=wd, w_sparse=ws)
        torch.cuda.synchronize()
        assert np.array_equal(fi.cpu().numpy(), ei) and np.array_equal(fs.cpu().numpy(), es)
        one = HipFlatIndex(d, "l2")
        one.add(x)
        ss, si = ShardedHybrid(one, HipBM25(full)).search_device(qd, sq, depth=depth, k=k, c=c, w_dense=wd, w_sparse=ws)
        torch.cuda.synchronize()
        assert np.array_equal(si.cpu().numpy(), ei) and np.array_equal(ss.cpu().numpy(), es)


def test_rrf_matches_oracle_bit_exact(gpu):
    from hiprag import rrf_fuse
    rng = np.random.default_rng(11)
    nq, depth = 33, 50
    a = np.stack([rng.permutation(400)[:depth] for _ in range(nq)]).astype(np.int64)
    b = np.stack([rng.permutation(400)[:depth] for _ in range(nq)]).astype(np.int64)
    a[0, 40:] = -1
    b[1, :] = -1
    a[2, :] = -1
    b[2, :] = -1
    b[3] = a[3]                                      # identical lists: every fused score ties pairwise by rank
    for (c, wa, wb, k) in [(60.0, 1.0, 1.0, 10), (60.0, 0.7, 0.3, 50), (1.0, 1.0, 1.0, 100)]:
        s, i = rrf_fuse(a, b, k, c=c, w_a=wa, w_b=wb)
        es, ei = ho.rrf_fuse(a, b, k, c=c, w_a=wa, w_b=wb)
        assert np.array_equal(i, ei)
        assert np.array_equal(s, es)


def test_hybrid_fast_path_one_call_and_library_bracket(gpu):
    """hiphybrid_search (dense top-depth + BM25 top-depth + RRF in one C-ABI call) against the oracle's three steps, and
    hiprag_init / hiprag_shutdown: shutdown drops every handle, the wrappers then report an unknown handle."""
    import hiprag
    from hiprag import HipBM25, HipFlatIndex, HipRagError, hybrid_search
    hiprag.init(1)
    with pytest.raises(HipRagError):
        hiprag.init(4096)                                            # more devices than any node has
    n, d, depth, k = 6000, 96, 50, 10
    x = ho.synthetic_vectors(n, d, seed=41)
    q = ho.synthetic_queries(9, d, seed=42)
    p = ho.synthetic_postings(n, n_terms=512, seed=43)
    sq = ho.synthetic_sparse_queries(9, n_terms=512, terms_per_query=5, seed=44, min_rank=4)
    sq[2] = np.asarray([], dtype=np.uint32)                          # a query without sparse terms: dense list only
    ix = HipFlatIndex(d, "ip")
    ix.add(x)
    bm = HipBM25(_gpu_postings(p))
    for (c, wd, ws) in [(60.0, 1.0, 1.0), (60.0, 0.7, 0.3)]:
        s, i = hybrid_search(ix, bm, q, sq, depth=depth, k=k, c=c, w_dense=wd, w_sparse=ws)
        _, di = ho.flat_search(x, q, depth, ho.METRIC_IP)
        _, bi = ho.bm25_search(p, sq, depth)
        es, ei = ho.rrf_fuse(di, bi, k, c=c, w_a=wd, w_b=ws)
        assert np.array_equal(i, ei) and np.array_equal(s, es)
    # shutdown drops EVERY handle of the process, so it is exercised in a process of its own
    import subprocess
    import sys
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import hiprag\n"
        "from hiprag import HipFlatIndex, HipRagError\n"
        "hiprag.init(1)\n"
        "x = np.random.default_rng(0).standard_normal((300, 64)).astype(np.float32)\n"
        "ix = HipFlatIndex(64, 'ip'); ix.add(x)\n"
        "assert ix.search(x[:2], 3)[1][:, 0].tolist() == [0, 1]\n"
        "hiprag.shutdown()\n"
        "try:\n"
        "    ix.search(x[:2], 3); raise SystemExit('handle survived the shutdown')\n"
        "except HipRagError:\n"
        "    pass\n"
        "ix._h = None\n"
        "ix2 = HipFlatIndex(64, 'ip'); ix2.add(x[:100])\n"
        "assert ix2.search(x[:1], 3)[1][0, 0] == 0\n"
        "print('bracket ok')\n") % (REPO, os.path.join(REPO, "intool-rag_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "bracket ok" in r.stdout, r.stdout + r.stderr


def test_overlay_search_matches_reference_golden(gpu, tmp_path, monkeypatch):
    """search_hip_by_vector == the reference's search_faiss_by_vector outputs (ids, order, enrichment, -1 quirk)."""
    import rag.storage.hip_index as hi
    g = GOLD["search_faiss_by_vector"]
    monkeypatch.setenv("STORAGE_DIR", str(tmp_path))
    monkeypatch.setenv("HIP_INDEX_METRIC", "l2")
    hi.clear_caches()
    x = np.asarray(g["vectors"], dtype=np.float32)
    index = hi.create_hip_index([list(map(float, row)) for row in x])           # list-of-lists like the reference
    hi.save_hip_index(index, str(tmp_path / f"{g['doc_id']}{hi.INDEX_SUFFIX}"))
    with open(tmp_path / f"{g['doc_id']}_chunks.json", "w") as f:
        json.dump(g["chunks_json"], f)
    hi.clear_caches()                                                              # force the load path
    asyncio.run(hi.initialize_storage())
    for case in g["cases"]:
        got = asyncio.run(hi.search_hip_by_vector(case["query"], limit=case["limit"]))
        exp = case["expected"]
        assert [r["chunk_id"] for r in got] == [r["chunk_id"] for r in exp]
        assert [{k: v for k, v in r.items() if k != "score"} for r in got] == \
               [{k: v for k, v in r.items() if k != "score"} for r in exp]
        assert np.allclose([r["score"] for r in got], [r["score"] for r in exp], rtol=0, atol=1e-4)
    reader = hi.HipIndexReader(str(tmp_path / f"{g['doc_id']}{hi.INDEX_SUFFIX}"))
    assert reader.get_dimension() == x.shape[1] and reader.get_size() == x.shape[0]
    hi.clear_caches()
    monkeypatch.setenv("STORAGE_DIR", str(tmp_path / "empty"))
    os.makedirs(tmp_path / "empty")
    assert asyncio.run(hi.search_hip_by_vector(case["query"], limit=5)) == g["no_index_expected"]


class _FakeProvider:
    """Stands for the encoder so the retriever can be tested without model weights."""

    def __init__(self, table):
        self.table = table

    async def embed_single(self, text, instruction=None):
        return [float(v) for v in self.table[text]]

    async def embed_batch(self, texts, instruction=None):
        return [[float(v) for v in self.table[t]] for t in texts]

    def dimension(self):
        return len(next(iter(self.table.values())))


def test_retriever_dense_and_hybrid_end_to_end(gpu, tmp_path, monkeypatch):
    import rag.storage.hip_index as hi
    from rag.storage.hip_index.sparse import clear_sparse_cache
    import rag.llm.embeddings.factory as fac
    from rag.query.retriever import HybridRetriever, retrieve_and_rank_pages
    monkeypatch.setenv("STORAGE_DIR", str(tmp_path))
    hi.clear_caches()
    clear_sparse_cache()
    rng = np.random.default_rng(3)
    n, d = 120, 64
    x = ho.synthetic_vectors(n, d, seed=41)
    words = ["invoice", "total", "tax", "customer", "address", "payment", "bank", "date", "item", "price"]
    texts = [" ".join(rng.choice(words, size=6)) + f" ref{i}" for i in range(n)]
    chunks = {"total": n, "chunks": [{"chunk_id": f"c_{1 + i // 5:03d}_{i % 5:03d}", "page": 1 + i // 5, "text": texts[i],
                                      "chunk_index": i % 5} for i in range(n)]}
    with open(tmp_path / "docB_chunks.json", "w") as f:
        json.dump(chunks, f)
    hi.save_hip_index(hi.create_hip_index(x), str(tmp_path / f"docB{hi.INDEX_SUFFIX}"))
    qtext = "bank payment ref17"
    qvec = x[17] + 0.1 * ho.synthetic_vectors(1, d, seed=42)[0]
    qvec /= np.linalg.norm(qvec)
    fac.set_embedding_provider(_FakeProvider({qtext: qvec}))
    try:
        # dense-only == the reference's pipeline restated by the oracle
        pages = asyncio.run(retrieve_and_rank_pages(qtext, top_pages=5))
        dist, ids = ho.flat_search(x, qvec.astype(np.float32), 50, ho.METRIC_L2)
        res = ho.reader_search_transform(dist[0], ids[0])
        oc = [ho.OChunk(chunks["chunks"][i]["chunk_id"], texts[i], s, chunks["chunks"][i]["page"]) for i, s in res]
        exp = ho.rank_pages(oc, 5)
        assert [(p.page, [c.chunk_id for c in p.chunks]) for p in pages] == [(p, cid) for p, _, cid in exp]
        assert np.allclose([p.score for p in pages], [s for _, s, _ in exp], atol=1e-4)
        assert pages[0].chunks[0].chunk_id == chunks["chunks"][17]["chunk_id"] or any(
            c.chunk_id == chunks["chunks"][17]["chunk_id"] for p in pages for c in p.chunks)
        # hybrid: dense top-50 + BM25 top-50 -> RRF, all three legs checked against the oracle
        r = HybridRetriever(top_chunks=50, top_pages=5, hybrid=True)
        got = asyncio.run(r.retrieve_chunks(qtext))
        op = ho.build_postings_from_texts(texts)
        qterms = [op.vocab[t] for t in ho.tokenize(qtext) if t in op.vocab]
        _, sparse_ids = ho.bm25_search(op, [qterms], 50)
        dense_ids = np.asarray([[i for i, _ in res]], dtype=np.int64)
        fs, fi = ho.rrf_fuse(dense_ids, sparse_ids, 50)
        exp_ids = [int(i) for i in fi[0] if i >= 0]
        assert [c.chunk_id for c in got] == [chunks["chunks"][i]["chunk_id"] for i in exp_ids]
        assert np.array_equal(np.asarray([c.metadata["rrf_score"] for c in got], np.float32), fs[0][:len(got)])
        assert got[0].chunk_id == chunks["chunks"][17]["chunk_id"]          # both legs agree on the planted chunk
    finally:
        fac.set_embedding_provider(None)
        hi.clear_caches()
        clear_sparse_cache()


def test_hip_embedding_provider_and_reranker_follow_reference_behaviour(gpu, monkeypatch):
    """EmbeddingProvider ABC semantics of rag/providers/hf/embeddings.py:42-88 on the GPU encoder, then rerank."""
    from hiprag import EncoderConfig, HipEncoder, random_state
    from oracle import encoder_oracle as eo
    from rag.llm.embeddings.base import EmbeddingProvider
    from rag.providers.hip.embeddings import HipEmbeddingProvider
    from rag.providers.hip.tokenizer import HashTokenizer
    from rag.query.reranker import CrossEncoderReranker
    from rag.query.retriever import RetrievedChunk
    cfg = EncoderConfig(vocab=2000, hidden=256, layers=2, heads=4, ffn=512, max_pos=200, max_seq_len=128)
    sd = random_state(cfg, seed=9, with_head=True)
    enc = HipEncoder(cfg, sd, with_head=True)
    tok = HashTokenizer(cfg.vocab)
    prov = HipEmbeddingProvider(encoder=enc, tokenizer=tok)
    assert isinstance(prov, EmbeddingProvider) and prov.dimension() == 256
    assert asyncio.run(prov.embed_single("")) == [0.0] * 256
    assert asyncio.run(prov.embed_single("   \n ")) == [0.0] * 256
    assert asyncio.run(prov.embed_batch([])) == []
    texts = ["  invoice total\namount ", None, "", "bank payment reference number 17"]
    vecs = asyncio.run(prov.embed_batch(texts, instruction="ignored", batch_size=3))
    assert len(vecs) == 4 and all(len(v) == 256 for v in vecs)
    single = asyncio.run(prov.embed_single(" invoice total\namount"))
    assert np.allclose(single, vecs[0], atol=2e-3)                    # strip + newline->space: same tokens
    assert np.allclose(vecs[1], vecs[2], atol=1e-6)                   # None and "" both encode the empty string
    assert abs(np.linalg.norm(vecs[1]) - 1.0) < 1e-3                  # ...and it is NOT the zero vector (hf :70-77)
    toks = [tok.encode(t, 128) for t in ["invoice total amount", "", "", "bank payment reference number 17"]]
    ref = eo.embed_fp32(eo.bf16_round_state(sd), toks, cfg.layers, cfg.heads)
    assert min(float(np.dot(vecs[i], ref[i])) for i in range(4)) >= 0.999
    # reranker: order by logit, ties by retrieval order, scores attached
    rr = CrossEncoderReranker(encoder=enc, tokenizer=tok, top_k=2)
    chunks = [RetrievedChunk(f"c{i}", t, 0.5, 1, {}) for i, t in
              enumerate(["bank payment", "tax invoice total", "date item price", "bank"])]
    out = asyncio.run(rr.rerank("bank payment ref", chunks))
    pairs = [tok.encode_pair("bank payment ref", c.text, 128) for c in chunks]
    ref_logits = eo.rerank_logits_fp32(eo.bf16_round_state(sd), pairs, cfg.layers, cfg.heads)
    got_logits = np.asarray(rr.score("bank payment ref", [c.text for c in chunks]))
    assert np.allclose(got_logits, ref_logits, atol=5e-2)
    assert [c.chunk_id for c in out] == [chunks[i].chunk_id for i in sorted(range(4), key=lambda i: (-got_logits[i], i))[:2]]
    assert all("rerank_score" in c.metadata for c in out)


def test_provider_loads_weights_and_tokenizer_from_local_files(gpu, tmp_path, monkeypatch):
    """HIP_ENCODER_WEIGHTS (safetensors, XLM-R parameter names -- here with the `roberta.` prefix checkpoints
    of sequence-classification heads carry) + HIP_TOKENIZER_FILE + HIP_ENCODER_CONFIG build the provider the factory
    returns; its embedding equals the encoder run directly on the same state and token ids."""
    import rag.llm.embeddings.factory as fac
    from safetensors.torch import save_file
    from hiprag import EncoderConfig, HipEncoder, random_state
    cfg = EncoderConfig(vocab=64, hidden=128, layers=2, heads=2, ffn=256, max_pos=40, max_seq_len=32)
    sd = random_state(cfg, seed=3)
    save_file({("roberta." + k): v.contiguous() for k, v in sd.items()}, str(tmp_path / "model.safetensors"))
    from tokenizers import Tokenizer
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import Whitespace
    tk = Tokenizer(WordLevel({"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3, "hello": 4, "world": 5, "bank": 6, "payment": 7,
                              "ref": 8}, unk_token="<unk>"))
    tk.pre_tokenizer = Whitespace()
    tk.save(str(tmp_path / "tokenizer.json"))
    monkeypatch.setenv("HIP_ENCODER_WEIGHTS", str(tmp_path / "model.safetensors"))
    monkeypatch.setenv("HIP_TOKENIZER_FILE", str(tmp_path / "tokenizer.json"))
    monkeypatch.setenv("HIP_ENCODER_CONFIG", json.dumps({"vocab": 64, "hidden": 128, "layers": 2, "heads": 2, "ffn": 256,
                                                         "max_pos": 40, "max_seq_len": 32}))
    monkeypatch.setenv("EMBEDDING_PROVIDER", "hip")
    fac.set_embedding_provider(None)
    try:
        prov = fac.get_embedding_provider()
        assert prov.dimension() == 128 and not prov.tokenizer.synthetic
        got = np.asarray(asyncio.run(prov.embed_single("hello world")), dtype=np.float32)
        ref = HipEncoder(cfg, sd).encode_tokens([[0, 4, 5, 2]]).cpu().numpy()[0]
        assert np.array_equal(got, ref)
        batch = np.asarray(asyncio.run(prov.embed_batch(["hello world", "", "bank payment ref"])), dtype=np.float32)
        assert np.array_equal(batch[0], ref) and batch.shape == (3, 128)
        # instruction prefixes: ignored like the reference by default, prepended with HIP_APPLY_INSTRUCTION
        import rag.providers.hip.embeddings as pe
        with_instr = np.asarray(asyncio.run(prov.embed_single("world", instruction="hello ")), dtype=np.float32)
        assert np.array_equal(with_instr, np.asarray(asyncio.run(prov.embed_single("world")), dtype=np.float32))
        monkeypatch.setattr(pe.config, "HIP_APPLY_INSTRUCTION", True, raising=False)
        assert np.array_equal(np.asarray(asyncio.run(prov.embed_single("world", instruction="hello ")), dtype=np.float32), ref)
        b2 = np.asarray(asyncio.run(prov.embed_batch(["world", ""], instruction="hello ")), dtype=np.float32)
        assert np.array_equal(b2[0], ref)
    finally:
        fac.set_embedding_provider(None)


def test_overlay_reads_index_files_written_by_the_reference_format(gpu, tmp_path, monkeypatch):
    """A STORAGE_DIR that only holds `{doc}_faiss.index` (the reference's own file) is searchable as is."""
    import rag.storage.hip_index as hi
    from hiprag.faiss_io import write_faiss_flat
    g = GOLD["search_faiss_by_vector"]
    monkeypatch.setenv("STORAGE_DIR", str(tmp_path))
    hi.clear_caches()
    x = np.asarray(g["vectors"], dtype=np.float32)
    write_faiss_flat(str(tmp_path / f"{g['doc_id']}_faiss.index"), x, 1)
    with open(tmp_path / f"{g['doc_id']}_chunks.json", "w") as f:
        json.dump(g["chunks_json"], f)
    asyncio.run(hi.initialize_storage())
    case = g["cases"][0]
    got = asyncio.run(hi.search_hip_by_vector(case["query"], limit=case["limit"]))
    assert [r["chunk_id"] for r in got] == [r["chunk_id"] for r in case["expected"]]
    hi.clear_caches()


def test_multi_document_search_merges_every_index(gpu, tmp_path, monkeypatch):
    """HIP_SEARCH_ALL_DOCUMENTS: every document's index is searched and the enriched rows merge by score;
    the default still answers from the first file only (faiss_index.py:162-167)."""
    import rag.storage.hip_index as hi
    from hiprag.faiss_io import write_faiss_flat
    from oracle import hybrid_oracle as ho
    monkeypatch.setenv("STORAGE_DIR", str(tmp_path))
    hi.clear_caches()
    rng = np.random.default_rng(99)
    docs = {"a_manual": 40, "b_short": 7, "c_report": 100}            # b_short is smaller than the limit: -1 padding
    rows = {}
    for n, (doc, cnt) in enumerate(docs.items()):
        x = rng.standard_normal((cnt, 64)).astype(np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        rows[doc] = x
        if doc == "c_report":                                          # one document still in the reference's format
            write_faiss_flat(str(tmp_path / f"{doc}_faiss.index"), x, 1)
        else:
            hi.save_hip_index(hi.create_hip_index(x, metric="l2"), str(tmp_path / f"{doc}{hi.INDEX_SUFFIX}"))
        chunks = [{"chunk_id": f"{doc}_{i}", "text": f"{doc} text {i}", "page": i // 4 + 1,
                   "metadata": {"title": doc, "source_filename": f"{doc}.pdf"}} for i in range(cnt)]
        with open(tmp_path / f"{doc}_chunks.json", "w") as f:
            json.dump({"chunks": chunks}, f)
    hi.clear_caches()
    q = rows["c_report"][17] + 0.3 * rows["a_manual"][5] + 0.2 * rows["b_short"][2]
    q = (q / np.linalg.norm(q)).astype(np.float32)
    limit = 10
    expected = []
    for di, doc in enumerate(sorted(docs)):
        d, i = ho.flat_search(rows[doc], q, limit, ho.METRIC_L2)
        for rank, (rid, score) in enumerate(r for r in ho.reader_search_transform(d[0], i[0]) if r[0] >= 0):
            expected.append((-score, di, rank, f"{doc}_{rid}"))
    expected.sort()
    monkeypatch.setattr(hi.config, "HIP_SEARCH_ALL_DOCUMENTS", True, raising=False)
    got = asyncio.run(hi.search_hip_by_vector([float(v) for v in q], limit=limit))
    assert [r["chunk_id"] for r in got] == [e[3] for e in expected[:limit]]
    assert np.allclose([r["score"] for r in got], [-e[0] for e in expected[:limit]], rtol=0, atol=1e-4)
    assert got[0]["chunk_id"] == "c_report_17" and {r["doc_id"] for r in got} >= {"a_manual", "c_report"}
    assert all(r["source_filename"] == r["doc_id"] + ".pdf" for r in got)
    monkeypatch.setattr(hi.config, "HIP_SEARCH_ALL_DOCUMENTS", False, raising=False)
    first = asyncio.run(hi.search_hip_by_vector([float(v) for v in q], limit=limit))
    assert len({r["chunk_id"].rsplit("_", 1)[0] for r in first}) == 1          # one document only, like the reference
    hi.clear_caches()


def test_ingest_indexing_then_retrieval_config1_scale(gpu, tmp_path, monkeypatch):
    """BASELINE configs[0] shape end to end on the GPU: 10k chunks -> index_chunks (embed_batch on the encoder, vectors
    straight from HBM into the index, postings from the same texts; rag/ingest/ingestion_pipeline.py:80-94) ->
    retrieve_and_rank_pages through the provider.  The retrieval legs are checked against the oracle ON THE VECTORS THE
    GPU STORED (the encoder's own parity is test_encoder_gpu.py); dense ids bit-exact, page ranking equal."""
    import rag.storage.hip_index as hi
    import rag.llm.embeddings.factory as fac
    from hiprag import EncoderConfig, HipEncoder, random_state
    from rag.ingest import index_chunks
    from rag.providers.hip.embeddings import HipEmbeddingProvider
    from rag.providers.hip.tokenizer import HashTokenizer
    from rag.query.retriever import HybridRetriever, retrieve_and_rank_pages
    from rag.storage.hip_index.sparse import clear_sparse_cache
    monkeypatch.setenv("STORAGE_DIR", str(tmp_path))
    hi.clear_caches()
    clear_sparse_cache()
    cfg = EncoderConfig(vocab=4000, hidden=128, layers=2, heads=2, ffn=256, max_pos=80, max_seq_len=64)
    enc = HipEncoder(cfg, random_state(cfg, seed=21))
    prov = HipEmbeddingProvider(encoder=enc, tokenizer=HashTokenizer(cfg.vocab))
    rng = np.random.default_rng(8)
    vocab = [f"w{i}" for i in range(600)]
    n = 10000
    texts = [" ".join(rng.choice(vocab, size=int(rng.integers(5, 14)))) + f" id{i}" for i in range(n)]
    chunks = {"total": n, "chunks": [{"chunk_id": f"c_{1 + i // 8:04d}_{i % 8:03d}", "page": 1 + i // 8, "text": texts[i],
                                      "chunk_index": i % 8} for i in range(n)]}
    with open(tmp_path / "docC_chunks.json", "w") as f:
        json.dump(chunks, f)
    fac.set_embedding_provider(prov)
    try:
        summary = asyncio.run(index_chunks("docC", chunks["chunks"], storage_dir=tmp_path, with_sparse=True))
        assert summary["success"] and summary["vectors_indexed"] == n and summary["postings_indexed"] > n
        assert (tmp_path / f"docC{hi.INDEX_SUFFIX}").exists()
        reader = hi.HipIndexReader(str(tmp_path / f"docC{hi.INDEX_SUFFIX}"))
        stored = np.stack([reader.index.reconstruct(i) for i in range(n)])
        assert np.allclose(np.linalg.norm(stored, axis=1), 1.0, atol=1e-3)
        for qtext in (texts[1234], texts[77] + " " + texts[78], "w3 w5 w8 id4321"):
            qvec = np.asarray(asyncio.run(prov.embed_single(qtext)), dtype=np.float32)
            pages = asyncio.run(retrieve_and_rank_pages(qtext, top_pages=5))
            dist, ids = ho.flat_search(stored, qvec, 50, ho.METRIC_L2)
            res = ho.reader_search_transform(dist[0], ids[0])
            oc = [ho.OChunk(chunks["chunks"][i]["chunk_id"], texts[i], s, chunks["chunks"][i]["page"]) for i, s in res]
            exp = ho.rank_pages(oc, 5)
            assert [(p.page, [c.chunk_id for c in p.chunks]) for p in pages] == [(p, cid) for p, _, cid in exp]
            # hybrid on top of the ingest-time postings
            got = asyncio.run(HybridRetriever(top_chunks=50, top_pages=5, hybrid=True).retrieve_chunks(qtext))
            op = ho.build_postings_from_texts(texts)
            qterms = [op.vocab[t] for t in ho.tokenize(qtext) if t in op.vocab]
            _, sparse_ids = ho.bm25_search(op, [qterms], 50)
            dense_ids = np.asarray([[i for i, _ in res]], dtype=np.int64)
            _, fi = ho.rrf_fuse(dense_ids, sparse_ids, 50)
            assert [c.chunk_id for c in got] == [chunks["chunks"][int(i)]["chunk_id"] for i in fi[0] if i >= 0]
        assert any(c.chunk_id == chunks["chunks"][1234]["chunk_id"] for p in asyncio.run(
            retrieve_and_rank_pages(texts[1234], top_pages=5)) for c in p.chunks)
    finally:
        fac.set_embedding_provider(None)
        hi.clear_caches()
        clear_sparse_cache()


def test_bm25_tiled_and_global_accumulator_forms_agree_with_the_oracle(gpu, monkeypatch):
    """Five document tiles (the last one partial), lists long enough for skip tables and short ones that every tile
    filters; k = 64 is the last k of the tiled form, k = 65 and a 70-term query take the global-accumulator form, which
    HIPBM25_GLOBAL_ACC=1 also forces for the first case.  All of them: ids and fp32 scores equal to the oracle's."""
    from hiprag import HipBM25, PostingsCSR, HipRagError
    n_docs, n_terms = 70001, 2048
    p = ho.synthetic_postings(n_docs, n_terms=n_terms, seed=31)
    df = np.diff(p.offsets)
    assert df.max() >= 2048 and (df[df > 0].min() < 2048)
    queries = ho.synthetic_sparse_queries(21, n_terms=n_terms, terms_per_query=6, seed=32, min_rank=4)
    queries[2] = np.arange(3, 73, dtype=np.uint32)                       # 70 terms
    queries[7] = np.asarray([0, 1, 2, 0], dtype=np.uint32)               # the three longest lists, one of them twice
    short = [q for i, q in enumerate(queries) if i != 2]
    for k in (10, 64, 65):
        es, ei = ho.bm25_search(p, short, k)
        s, i = HipBM25(_gpu_postings(p)).search(short, k)
        assert np.array_equal(i, ei) and np.array_equal(s, es), k
    es, ei = ho.bm25_search(p, queries, 10)
    s, i = HipBM25(_gpu_postings(p)).search(queries, 10)                 # the 70-term query sends the batch the other way
    assert np.array_equal(i, ei) and np.array_equal(s, es)
    monkeypatch.setenv("HIPBM25_GLOBAL_ACC", "1")
    s, i = HipBM25(_gpu_postings(p)).search(short, 10)
    es, ei = ho.bm25_search(p, short, 10)
    assert np.array_equal(i, ei) and np.array_equal(s, es)
    monkeypatch.delenv("HIPBM25_GLOBAL_ACC")
    # lists must be strictly ascending by document: the skip tables and the atomics-free accumulation rely on it
    bad = p.doc_ids.copy()
    t = int(np.argmax(df))
    lo = int(p.offsets[t])
    bad[lo], bad[lo + 1] = bad[lo + 1], bad[lo]
    with pytest.raises(HipRagError, match="ascending"):
        HipBM25(PostingsCSR(p.n_docs, p.n_terms, p.offsets, bad, p.impacts))


def test_bm25_tiled_form_beyond_three_million_documents(gpu):
    """4.2 M documents at depth 50 = 456 tiles x 4 lists x 50 = 91 k candidate slots per query, past the 65536 the tiled form
    used to stop at (a 10M-document shard on one GPU then took the global-accumulator form: 200 ms per 256 queries): the
    lists of every length go through merge_packed_loop_kernel.  Short documents (three tokens each, 1000 terms) keep the
    oracle cheap; ids and fp32 scores bit-exact, also for a query of one frequent term and one of an unknown term."""
    from hiprag import HipBM25
    n_docs, n_terms, depth = 4_200_000, 1000, 50
    rng = np.random.default_rng(61)
    doc = np.repeat(np.arange(n_docs, dtype=np.int64), 3)
    w = 1.0 / np.arange(1, n_terms + 1)
    cdf = np.cumsum(w) / w.sum()
    term = np.minimum(np.searchsorted(cdf, rng.random(doc.size)), n_terms - 1)
    p = ho.build_postings_from_pairs(doc, term, n_docs, n_terms)
    bm = HipBM25(_gpu_postings(p))
    sq = [rng.choice(np.arange(20, n_terms), size=5, replace=False).astype(np.uint32) for _ in range(10)]
    sq += [np.asarray([0], dtype=np.uint32), np.asarray([3, 5000], dtype=np.uint32)]
    es, ei = ho.bm25_search(p, sq, depth)
    gs, gi = bm.search(sq, depth)
    assert np.array_equal(gi, ei) and np.array_equal(gs, es)
    st = bm.stats()
    assert st["queries"] >= len(sq)


def test_hybrid_search_device_overlaps_the_legs_and_equals_the_oracle(gpu):
    """hiphybrid_search_dev (hiprag.hybrid_search_device: the dense leg on the library's high-priority scan stream with
    CUs left out of its grid, BM25 beside it on the caller's stream, the fusion behind both) and the host-array call
    hiphybrid_search around it: fused lists bit-exact vs the oracle call after call, from the default stream and from a
    stream of the caller's; the per-leg lists the call leaves behind equal the oracle's; the index's own spare-CU
    setting survives the call."""
    import torch
    from hiprag import HipBM25, HipFlatIndex, hybrid_search, hybrid_search_device
    n, d, depth, k, nq = 30000, 128, 50, 10, 70
    x = ho.synthetic_vectors(n, d, seed=71)
    q = ho.synthetic_queries(nq, d, seed=72)
    p = ho.synthetic_postings(n, n_terms=1024, seed=73)
    sq = ho.synthetic_sparse_queries(nq, n_terms=1024, terms_per_query=5, seed=74, min_rank=4)
    ix = HipFlatIndex(d, "ip")
    ix.add(x)
    bm = HipBM25(_gpu_postings(p))
    _, di = ho.flat_search(x, q, depth, ho.METRIC_IP)
    bs, bi = ho.bm25_search(p, sq, depth)
    es, ei = ho.rrf_fuse(di, bi, k)
    qd = torch.from_numpy(q).cuda()
    ix.set_spare_cus(3)
    for _ in range(4):
        fs, fi = hybrid_search_device(ix, bm, qd, sq, depth=depth, k=k)
        torch.cuda.synchronize()
        assert np.array_equal(fi.cpu().numpy(), ei) and np.array_equal(fs.cpu().numpy(), es)
        hs, hi = hybrid_search(ix, bm, q, sq, depth=depth, k=k)
        assert np.array_equal(hi, ei) and np.array_equal(hs, es)
    assert ix.spare_cus == 3
    mine = torch.cuda.Stream()
    with torch.cuda.stream(mine):
        q2 = qd * 1.0                                  # produced on the caller's stream right before the call
        fs, fi, ((d64, dids), (b64, bids)) = hybrid_search_device(ix, bm, q2, sq, depth=depth, k=k, return_lists=True)
        total = fs.sum()                               # consumed on it right after
    mine.synchronize()
    assert np.array_equal(fi.cpu().numpy(), ei) and np.array_equal(fs.cpu().numpy(), es) and np.isfinite(float(total))
    assert np.array_equal(dids.cpu().numpy(), di) and np.array_equal(bids.cpu().numpy(), bi)
    assert np.array_equal(b64.cpu().numpy().astype(np.float32), bs) and d64.shape == (nq, depth)
    assert hybrid_search_device(ix, bm, qd[:0], [], depth=depth, k=k)[1].shape == (0, k)
