#!/usr/bin/env python3
"""
End-to-end shape of BASELINE config 5 on ONE GPU: embed a batch of token sequences with the BGE-M3-shaped encoder ->
hybrid retrieve (dense top-50 + BM25 top-50 + RRF) over a synthetic corpus -> cross-encoder re-rank of the fused top-k.
Random weights, synthetic tokens / postings (no checkpoint or tokenizer offline): this measures the pipeline's cost, not
retrieval quality.  Prints one JSON line with per-stage times -- after checking every stage against the CPU oracle fed the
GPU's output of the stage before it (a sample of the embeddings and of the rerank logits against the fp32 oracle within the
tolerances tests/test_configs_gpu.py measured; dense, BM25 and fused lists of a sample of queries bit-exact), the way
test_config4_shape_embed_hybrid_rerank_stagewise does; a stage that disagrees aborts the run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "intool-rag_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=200_000)
    ap.add_argument("--nq", type=int, default=256)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--rerank_queries", type=int, default=16, help="queries whose fused top-k is re-ranked")
    ap.add_argument("--k", type=int, default=10)
    args = ap.parse_args()
    import torch
    from hiprag import EncoderConfig, HipBM25, HipEncoder, HipFlatIndex, build_postings, rrf_fuse_device
    from oracle import hybrid_oracle as ho
    dev = torch.device("cuda", 0)
    cfg = EncoderConfig(layers=args.layers)
    from hiprag import random_state
    from oracle import encoder_oracle as eo
    sd_e, sd_r = random_state(cfg, seed=0), random_state(cfg, seed=1, with_head=True)
    enc = HipEncoder(cfg, sd_e)
    rer = HipEncoder(cfg, sd_r)
    rng = np.random.default_rng(0)
    N = args.docs
    index = HipFlatIndex(cfg.hidden, "ip")
    host_rows = []
    for c in range(0, N, 100000):
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + c)
        x = torch.randn((min(100000, N - c), cfg.hidden), generator=g, device=dev)
        x /= x.norm(dim=1, keepdim=True)
        index.add_device(x)
        host_rows.append(x.cpu().numpy())
    host_rows = np.concatenate(host_rows)
    p = ho.synthetic_postings(min(N, 200_000), n_terms=65536, seed=777)
    if p.n_docs != N:
        raise SystemExit("--docs above 200000 needs the torch postings generator of bench_hybrid.py")
    from hiprag import PostingsCSR
    bm25 = HipBM25(PostingsCSR(p.n_docs, p.n_terms, p.offsets, p.doc_ids, p.impacts))
    sparse_q = ho.synthetic_sparse_queries(args.nq, n_terms=65536, seed=888)
    toks = [[0] + rng.integers(3, cfg.vocab, size=args.seq - 2).tolist() + [2] for _ in range(args.nq)]

    def sync():
        torch.cuda.synchronize()

    enc.encode_tokens(toks[:8])
    sync()
    t0 = time.perf_counter()
    qv = enc.encode_tokens(toks, batch_size=args.nq)
    sync()
    t_embed = time.perf_counter() - t0
    from hiprag import hybrid_search_device
    hybrid_search_device(index, bm25, qv[:8], sparse_q[:8], depth=50, k=args.k)      # workspaces and staging exist
    sync()
    t0 = time.perf_counter()
    fs, fi, ((d64, dids), (s64, sids)) = hybrid_search_device(index, bm25, qv, sparse_q, depth=50, k=args.k, return_lists=True)
    sync()
    t_retr = time.perf_counter() - t0
    d, s = (d64, None, dids), (s64, s64.to(torch.float32), sids)
    nr = min(args.rerank_queries, args.nq)
    fused = fi[:nr].cpu().numpy()
    pairs = []
    for qi in range(nr):
        qtok = toks[qi][1:65]                      # a 64-token query segment
        for _doc in fused[qi]:
            passage = rng.integers(3, cfg.vocab, size=args.seq - len(qtok) - 4).tolist()
            pairs.append([0] + qtok + [2, 2] + passage + [2])
    rer.score_tokens(pairs[:4])
    sync()
    t0 = time.perf_counter()
    logits = rer.score_tokens(pairs, batch_size=128)
    sync()
    t_rerank = time.perf_counter() - t0
    # ---- every stage against the oracle, fed the GPU's output of the stage before ----------------------------------------
    checks = {}
    pick = [0, args.nq // 2, args.nq - 1]
    ref = eo.embed_fp32(eo.bf16_round_state(sd_e), [toks[i] for i in pick], cfg.layers, cfg.heads, cfg.pad_id, cfg.ln_eps)
    got = qv[pick].cpu().numpy()
    checks["embed_min_cosine"] = round(float(np.min(np.sum(got * ref, axis=1))), 6)
    checks["embed_max_abs_delta"] = float(np.abs(got - ref).max())
    assert checks["embed_min_cosine"] >= 0.9998 and checks["embed_max_abs_delta"] <= 0.1 / np.sqrt(cfg.hidden), checks
    qs = list(range(0, args.nq, max(1, args.nq // 8)))[:8]
    qh = qv.cpu().numpy()
    _, di = ho.flat_search(host_rows, qh[qs], 50, ho.METRIC_IP)
    bs_, bi = ho.bm25_search(p, [sparse_q[i] for i in qs], 50)
    efs, efi = ho.rrf_fuse(di, bi, args.k)
    checks["dense_ids_equal"] = bool(np.array_equal(d[2][qs].cpu().numpy(), di))
    checks["bm25_ids_scores_equal"] = bool(np.array_equal(s[2][qs].cpu().numpy(), bi) and np.array_equal(s[1][qs].cpu().numpy(), bs_))
    checks["fused_equal"] = bool(np.array_equal(fi[qs].cpu().numpy(), efi) and np.array_equal(fs[qs].cpu().numpy(), efs))
    assert checks["dense_ids_equal"] and checks["bm25_ids_scores_equal"] and checks["fused_equal"], checks
    want = eo.rerank_logits_fp32(eo.bf16_round_state(sd_r), pairs[:3], cfg.layers, cfg.heads, cfg.pad_id, cfg.ln_eps)
    checks["rerank_max_abs_logit_delta"] = float(np.abs(logits[:3].cpu().numpy() - want).max())
    assert checks["rerank_max_abs_logit_delta"] <= 1.5e-2, checks
    print(json.dumps({
        "checks_against_the_oracle": checks,
        "workload": f"config 5 shape on 1 GPU: embed {args.nq}x{args.seq} tokens -> hybrid top-50/50 -> RRF top-{args.k} over "
                    f"{N} chunks -> rerank {nr}x{args.k} pairs of {args.seq} tokens",
        "embed_ms": round(t_embed * 1e3, 1), "embed_tokens_per_s": round(args.nq * args.seq / t_embed),
        "retrieve_ms": round(t_retr * 1e3, 2), "rerank_ms": round(t_rerank * 1e3, 1),
        "rerank_pairs_per_s": round(len(pairs) / t_rerank, 1), "logits_finite": bool(torch.isfinite(logits).all().item())}))


if __name__ == "__main__":
    main()
