"""
GPU parity of the IVF-Flat layer (libhiprag hipivf_*): at nprobe = nlist every row is scored and the result must equal the flat
search's -- ids bit-exact against the CPU oracle, scores within 1e-4 (they are the same fp64 re-scores); at small nprobe the
result is approximate but never wrong: every returned (id, score) is that row's exact score, the list is in canonical order,
and recall is what a coarse quantiser buys.  The reference builds faiss.IndexFlatL2 only (rag/storage/faiss_index.py:123).
"""
import numpy as np
import pytest

from oracle import hybrid_oracle as ho

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("metric", [ho.METRIC_IP, ho.METRIC_L2])
@pytest.mark.parametrize("n,d,nlist,k", [(6000, 256, 16, 10), (20011, 128, 64, 50), (700, 64, 7, 5)])
def test_full_probe_equals_the_flat_search(gpu, metric, n, d, nlist, k):
    from hiprag import HipIVFIndex
    x = ho.synthetic_vectors(n, d, seed=31)
    x[40:60] = x[3]                                  # exact duplicates: ties must come out by original id
    q = ho.synthetic_queries(23, d, seed=32)
    q[2] = x[3]
    ix = HipIVFIndex(d, nlist, metric)
    ix.train_add(x, iters=3)
    assert ix.ntotal == n and int(ix.list_lengths.sum()) == n
    s, i = ix.search(q, k, nprobe=nlist)
    es, ei = ho.flat_search(x, q, k, metric)
    assert np.array_equal(i, ei)
    assert np.allclose(s, es, rtol=0, atol=1e-4)
    s2, i2 = ix.search(q, k, nprobe=nlist + 5)       # more probes than lists: the same
    assert np.array_equal(i2, ei)


def test_small_probe_is_approximate_but_never_wrong(gpu):
    from hiprag import HipIVFIndex
    n, d, nlist, k = 50000, 128, 128, 10
    x = ho.synthetic_vectors(n, d, seed=41)
    rng = np.random.default_rng(42)
    q = x[rng.integers(0, n, size=64)] + 0.05 * rng.standard_normal((64, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    ix = HipIVFIndex(d, nlist, "ip")
    ix.train_add(x, iters=4)
    es, ei = ho.flat_search(x, q, k, ho.METRIC_IP)
    last = 0.0
    for nprobe in (1, 8, 32, 128):
        s, i = ix.search(q, k, nprobe)
        valid = i >= 0
        exact = np.einsum("qkd,qd->qk", x[np.where(valid, i, 0)].astype(np.float64), q.astype(np.float64))
        assert np.allclose(s[valid], exact[valid], rtol=0, atol=1e-4)            # every score is that row's exact score
        assert np.all((s[:, :-1] > s[:, 1:]) | ((s[:, :-1] == s[:, 1:]) & (i[:, :-1] < i[:, 1:])) | ~valid[:, 1:])
        recall = np.mean([len(set(a) & set(b)) / k for a, b in zip(i, ei)])
        assert recall >= last - 1e-9                                             # more probes never lose a hit
        last = recall
        if nprobe == 1:
            assert np.mean(i[:, 0] == ei[:, 0]) >= 0.5                           # a planted neighbour sits in the query's own list
    assert last == 1.0                                                           # nprobe = nlist: the flat result


def test_bad_arguments_raise(gpu):
    import torch
    from hiprag import HipIVFIndex, HipRagError
    x = ho.synthetic_vectors(500, 32, seed=5)
    ix = HipIVFIndex(32, 8, "l2")
    with pytest.raises(RuntimeError):
        ix.search(x[:1], 5, 2)                       # not built
    with pytest.raises(ValueError):
        ix.train_add(x[:4])                          # fewer rows than lists
    ix.train_add(x)
    with pytest.raises(HipRagError):
        ix.search_device(torch.from_numpy(x[:2]).cuda(), 300, 2)     # k beyond the probe kernel's list
    with pytest.raises(HipRagError):
        ix.search_device(torch.from_numpy(x[:2]).cuda(), 5, 0)
