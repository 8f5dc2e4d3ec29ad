"""
The dense finish's selector (hiprag_select_topk_dev = select_threshold_topk, csrc/topk_device.h) on plain arrays against
numpy: exact top-k under (value descending, index ascending), -inf = absent.  The cases are the ones the threshold design
could get wrong: counts above the candidate list (constant arrays, winners concentrated in few threads' strided sets ->
the bisection on packed keys), fewer values than k, ragged lengths, all absent, array lengths far above one workgroup's
single pass.  The reference has no selector of its own (FAISS's heap inside IndexFlat::search, rag/storage/faiss_index.py:137).
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _expected(vals, k):
    n_rows, n = vals.shape
    ev = np.full((n_rows, k), -np.inf, dtype=np.float32)
    ei = np.full((n_rows, k), -1, dtype=np.int64)
    for r in range(n_rows):
        v = vals[r]
        ok = np.flatnonzero(v != -np.inf)
        order = ok[np.lexsort((ok, -v[ok].astype(np.float64)))][:k]   # value desc, then index asc
        ev[r, :len(order)] = v[order]
        ei[r, :len(order)] = order
    return ev, ei


def _select(vals, k):
    import torch
    from hiprag import _native as nat
    n_rows, n = vals.shape
    stride = (n + 3) // 4 * 4
    buf = torch.full((n_rows, stride), float("nan"), dtype=torch.float32, device="cuda")   # padding must never be read as data
    buf[:, :n] = torch.from_numpy(vals).cuda()
    ov = torch.empty((n_rows, k), dtype=torch.float32, device="cuda")
    oi = torch.empty((n_rows, k), dtype=torch.int64, device="cuda")
    nat.call("hiprag_select_topk_dev", buf.data_ptr(), stride, n, n_rows, k, ov.data_ptr(), oi.data_ptr(),
             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return ov.cpu().numpy(), oi.cpu().numpy()


def _check(vals, k):
    gv, gi = _select(vals, k)
    ev, ei = _expected(vals, k)
    assert np.array_equal(gi, ei), f"indices differ in rows {np.flatnonzero((gi != ei).any(1))[:5]}"
    assert np.array_equal(gv, ev)


@pytest.mark.parametrize("n", [1, 3, 4, 5, 63, 64, 4095, 4096, 4097, 62500, 65536, 200003])
@pytest.mark.parametrize("k", [1, 27, 64])
def test_random_values_ragged_lengths(gpu, n, k):
    rng = np.random.default_rng(n * 131 + k)
    _check(rng.standard_normal((3, n)).astype(np.float32), k)


@pytest.mark.parametrize("k", [1, 27, 64])
def test_constant_and_few_valued_arrays(gpu, k):
    n = 70001
    rng = np.random.default_rng(k)
    vals = np.stack([np.full(n, 0.25, np.float32),                      # every value equal: the k lowest indices win
                     np.zeros(n, np.float32),
                     rng.integers(0, 3, n).astype(np.float32),           # three distinct values, ~23k ties at the top
                     -np.abs(rng.integers(0, 2, n)).astype(np.float32)])
    _check(vals, k)


def test_winners_concentrated_in_a_few_threads_take_the_bisection(gpu):
    # thread t of the selector owns indices {4 (1024 i + t) + 0..3}: put ALL large values into the sets of 20 threads, so
    # fewer than k thread maxima are large, the threshold falls to the background level and the first count is the whole
    # array's background share above it -- far beyond the 256-entry list
    n = 262144
    rng = np.random.default_rng(7)
    vals = rng.uniform(0.0, 1.0, (2, n)).astype(np.float32)
    idx = np.arange(n)
    owner = (idx // 4) % 1024
    hot = np.isin(owner, rng.choice(1024, 20, replace=False))
    vals[0, hot] += 10.0
    vals[1, hot] = 5.0          # and the same with ties among all of the winners
    for k in (27, 64):
        _check(vals, k)


def test_absent_values_and_rows_with_fewer_than_k(gpu):
    n = 10000
    rng = np.random.default_rng(3)
    vals = np.full((4, n), -np.inf, dtype=np.float32)
    vals[1, rng.choice(n, 5, replace=False)] = rng.standard_normal(5).astype(np.float32)      # 5 values, k = 27
    vals[2, :] = -np.finfo(np.float32).max                                                     # the dense scan's padding value
    vals[3, rng.choice(n, 300, replace=False)] = 1.0
    _check(vals, 27)


def test_signed_zeros_and_denormals_order_like_the_packed_keys(gpu):
    # ord32 orders -0.0 below +0.0: the expected order is the integer image's, so build it from the bit patterns
    vals = np.array([[0.0, -0.0, 1e-45, -1e-45, 1e-38, -1e-38, 0.0, -0.0]], dtype=np.float32)
    gv, gi = _select(vals, 8)
    bits = vals[0].view(np.uint32).astype(np.int64)
    img = np.where(bits & 0x80000000, 0xFFFFFFFF - bits, bits + 0x80000000)
    order = np.lexsort((np.arange(8), -img))
    assert np.array_equal(gi[0], order)
    assert np.array_equal(gv[0].view(np.uint32), vals[0][order].view(np.uint32))


def test_many_rows_at_the_dense_shape(gpu):
    rng = np.random.default_rng(11)
    _check(rng.standard_normal((96, 62500)).astype(np.float32), 27)


def test_rejects_bad_shapes(gpu):
    import torch
    from hiprag import _native as nat
    buf = torch.zeros(64, device="cuda")
    out_v = torch.zeros(65, device="cuda")
    out_i = torch.zeros(65, dtype=torch.int64, device="cuda")
    with pytest.raises(nat.HipRagError):
        nat.call("hiprag_select_topk_dev", buf.data_ptr(), 64, 64, 1, 65, out_v.data_ptr(), out_i.data_ptr(), None)
    with pytest.raises(nat.HipRagError):
        nat.call("hiprag_select_topk_dev", buf.data_ptr(), 62, 62, 1, 4, out_v.data_ptr(), out_i.data_ptr(), None)
