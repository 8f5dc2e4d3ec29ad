"""Long randomised parity soak of BM25 against the oracle (not collected by pytest; `python tests/soak_bm25_gpu.py` on a GPU
box): synthetic Zipf postings over several collection sizes / k / query lengths, and collections of a few repeated texts
(thousands of exact ties across document tiles)."""
import os, sys, numpy as np
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'intool-rag_amd'))
from oracle import hybrid_oracle as ho
from hiprag import HipBM25, PostingsCSR, build_postings_from_texts
rng=np.random.default_rng(99); bad=0
for case in range(60):
    n_docs=int(rng.choice([50, 300, 5000, 16384, 16385, 40000, 70000])); n_terms=int(rng.choice([64, 512, 4096]))
    k=int(rng.choice([1,7,10,50,63,64]))
    p=ho.synthetic_postings(n_docs, n_terms=n_terms, seed=100+case)
    tq=int(rng.choice([1,3,6,12]))
    queries=ho.synthetic_sparse_queries(int(rng.choice([1,5,33])), n_terms=n_terms, terms_per_query=min(tq,n_terms//2), seed=200+case, min_rank=int(rng.choice([0,4,16])) if n_terms>64 else 0)
    es,ei=ho.bm25_search(p,queries,k)
    ix=HipBM25(PostingsCSR(p.n_docs,p.n_terms,p.offsets,p.doc_ids,p.impacts))
    s,i=ix.search(queries,k)
    if not (np.array_equal(i,ei) and np.array_equal(s,es)):
        bad+=1; print("FAIL synthetic", case, n_docs, n_terms, k, flush=True)
# heavy ties: few distinct texts repeated over several tiles
for case in range(12):
    words=[f"w{j}" for j in range(12)]
    base=[" ".join(rng.choice(words,size=int(rng.integers(1,5)))) for _ in range(int(rng.choice([3,9,40])))]
    n=int(rng.choice([2000, 20000, 50000]))
    texts=[base[int(t)] for t in rng.integers(0,len(base),size=n)]
    p=build_postings_from_texts(texts); op=ho.build_postings_from_texts(texts)
    ix=HipBM25(p)
    q=[ix.terms_of(" ".join(rng.choice(words,size=3))) for _ in range(6)]
    for k in (5, 50, 64):
        s,i=ix.search(q,k); es,ei=ho.bm25_search(op,q,k)
        if not (np.array_equal(i,ei) and np.array_equal(s,es)):
            bad+=1; print("FAIL ties", case, n, k, flush=True)
print("bm25 soak done, failures:", bad)
