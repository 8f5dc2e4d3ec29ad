"""
CPU: the oracle against (1) the golden vectors generated from the reference's own Python
(tests/golden/make_reference_golden.py) and (2) hand-computed known answers.  No GPU, no reference at run time.
"""
import json
import os

import numpy as np
import pytest

from oracle import hybrid_oracle as ho

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_wrapper_golden.json")))


def test_reader_search_transform_matches_reference():
    for case in GOLD["reader_search"]:
        got = ho.reader_search_transform(case["distances_f32"], case["ids"])
        assert [[a, b] for a, b in got] == case["expected"]


def test_enrichment_matches_reference_including_minus_one_quirk():
    g = GOLD["search_faiss_by_vector"]
    x = np.asarray(g["vectors"], dtype=np.float32)
    # chunk list order = dict(values) order of load_chunks = file order with duplicate ids collapsed
    chunks = {c["chunk_id"]: c for c in g["chunks_json"]["chunks"]}
    assert list(chunks.keys()) == g["loaded_chunk_ids_in_order"]
    chunks_list = list(chunks.values())
    for case in g["cases"]:
        q = np.asarray(case["query"], dtype=np.float32)
        d, i = ho.flat_search(x, q, case["limit"], ho.METRIC_L2)
        res = ho.reader_search_transform(d[0], i[0])
        got = ho.enrich_results(res, chunks_list)
        assert got == case["expected"]
    assert g["no_index_expected"] == []


def test_page_ranking_matches_reference():
    for case in GOLD["page_ranking"]:
        chunks = [ho.OChunk(f"c{i}", f"t{i}", s, p, {"title": f"T{p}"}) for i, (s, p) in enumerate(case["chunks"])]
        got = ho.rank_pages(chunks, case["max_pages"])
        exp = case["expected"]
        assert [(p, s, ids) for p, s, ids in got] == [(e["page"], e["score"], e["chunk_ids"]) for e in exp]


def test_flat_search_known_answers():
    x = np.asarray([[1, 0], [0, 1], [1, 1], [-1, 0], [1, 0]], dtype=np.float32)
    q = np.asarray([[1, 0]], dtype=np.float32)
    s, i = ho.flat_search(x, q, 4, ho.METRIC_IP)
    assert i.tolist() == [[0, 2, 4, 1]] and s.tolist() == [[1.0, 1.0, 1.0, 0.0]]      # ties -> lower id first
    s, i = ho.flat_search(x, q, 6, ho.METRIC_L2)
    assert i.tolist() == [[0, 4, 2, 1, 3, -1]]
    assert s[0, :5].tolist() == [0.0, 0.0, 1.0, 2.0, 4.0] and s[0, 5] == np.finfo(np.float32).max
    s, i = ho.flat_search(x, q, 2, ho.METRIC_IP, id_base=100)
    assert i.tolist() == [[100, 102]]


def test_f32_twin_agrees_with_truth_on_separated_data():
    x = ho.synthetic_vectors(2000, 64, seed=5)
    q = ho.synthetic_queries(5, 64, seed=6)
    for metric in (ho.METRIC_IP, ho.METRIC_L2):
        s64, i64_ = ho.flat_search(x, q, 10, metric)
        s32, i32 = ho.flat_search_f32_faithful(x, q, 10, metric)
        assert np.array_equal(i64_, i32)
        assert np.allclose(s64, s32, atol=1e-5)


def test_merge_partial_topk_equals_unsharded():
    x = ho.synthetic_vectors(3000, 32, seed=7)
    q = ho.synthetic_queries(4, 32, seed=8)
    for metric in (ho.METRIC_IP, ho.METRIC_L2):
        fs, fi, f64 = ho.flat_search(x, q, 10, metric, return_f64=True)
        parts = [(0, 700), (700, 701), (701, 3000)]
        ps, pi = [], []
        for lo, hi in parts:
            _, ids, s64 = ho.flat_search(x[lo:hi], q, 10, metric, id_base=lo, return_f64=True)
            ps.append(s64)
            pi.append(ids)
        ms, mi = ho.merge_partial_topk(ps, pi, 10, metric)
        assert np.array_equal(mi, fi) and np.array_equal(ms, f64)


def test_bm25_known_answer():
    # 3 docs, hand-computed: N=3, avgdl = (3+2+4)/3 = 3
    texts = ["red apple pie", "apple tart", "Red red wine list"]
    p = ho.build_postings_from_texts(texts)
    assert p.vocab["red"] == 0 and p.n_docs == 3
    k1, b = ho.BM25_K1, ho.BM25_B
    idf_red = np.log(1 + (3 - 2 + 0.5) / (2 + 0.5))
    imp_d2 = idf_red * 2 * (k1 + 1) / (2 + k1 * (1 - b + b * 4 / 3))
    imp_d0 = idf_red * 1 * (k1 + 1) / (1 + k1 * (1 - b + b * 3 / 3))
    acc = ho.bm25_scores_taat(p, [p.vocab["red"]])
    assert acc[2] == np.float32(imp_d2) and acc[0] == np.float32(imp_d0) and acc[1] == 0
    s, i = ho.bm25_search(p, [[p.vocab["red"], p.vocab["apple"]]], 3)
    assert i[0].tolist()[:3] == [0, 2, 1]          # doc0 has both terms
    s, i = ho.bm25_search(p, [[p.vocab["wine"]]], 3)
    assert i[0].tolist() == [2, -1, -1]            # zero-score docs are excluded


def test_bm25_term_order_is_the_sum_order():
    p = ho.synthetic_postings(300, n_terms=64, seed=3)
    q = [5, 9, 2, 40]
    a = ho.bm25_scores_taat(p, q)
    acc = np.zeros(p.n_docs, np.float32)
    for t in q:
        lo, hi = int(p.offsets[t]), int(p.offsets[t + 1])
        for j in range(lo, hi):
            acc[p.doc_ids[j]] = np.float32(acc[p.doc_ids[j]] + p.impacts[j])
    assert np.array_equal(a, acc)


def test_rrf_known_answer():
    a = np.asarray([[10, 20, 30, -1]])
    b = np.asarray([[30, 40, 10, 50]])
    s, i = ho.rrf_fuse(a, b, 5)
    f = np.float32
    exp = {10: f(1) / f(61) + f(1) / f(63), 30: f(1) / f(63) + f(1) / f(61), 20: f(1) / f(62), 40: f(1) / f(62),
           50: f(1) / f(64)}
    assert i[0].tolist() == [10, 30, 20, 40, 50]                      # 10 vs 30 and 20 vs 40 tie -> lower id first
    assert [np.float32(v) for v in s[0]] == [f(exp[d]) for d in i[0]]
    s, i = ho.rrf_fuse(a, b, 2, w_a=0.7, w_b=0.3)
    assert i[0].tolist() == [10, 30]
    s, i = ho.rrf_fuse(np.asarray([[-1, -1]]), np.asarray([[-1]]), 3)
    assert i[0].tolist() == [-1, -1, -1]
