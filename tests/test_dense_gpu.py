"""
GPU parity of the dense flat index (libhiprag hipidx_*) against the CPU oracle (fp64 truth, (score, id) order).
Bar: ids bit-exact, scores within 1e-4 (they are fp64-rescored, so in practice within 1 ulp of fp32).
Mirrors what the reference's path does: IndexFlatL2 build + search (rag/storage/faiss_index.py:81-83,121-124).
"""
import numpy as np
import pytest

from oracle import hybrid_oracle as ho

pytestmark = pytest.mark.gpu
TOL = 1e-4
FLT_MAX = np.finfo(np.float32).max


def _check(index, x, q, k, metric, id_base=0):
    s, i = index.search(q, k)
    es, ei = ho.flat_search(x, q, k, metric, id_base=id_base)
    assert np.array_equal(i, ei), f"ids differ: first bad query {np.argwhere((i != ei).any(1))[:3].ravel()}"
    valid = ei >= 0
    assert np.allclose(s[valid], es[valid], rtol=0, atol=TOL)
    pad = FLT_MAX if metric == ho.METRIC_L2 else -FLT_MAX
    assert np.all(s[~valid] == pad)
    return s, i


@pytest.mark.parametrize("metric", [ho.METRIC_IP, ho.METRIC_L2])
@pytest.mark.parametrize("n,d,nq,k", [(5000, 1024, 7, 10), (4097, 384, 33, 10), (1000, 768, 1, 50), (64, 16, 3, 5),
                                      (31, 100, 2, 4), (20000, 1024, 32, 10)])
def test_parity_random_unit_vectors(gpu, metric, n, d, nq, k):
    from hiprag import HipFlatIndex
    x = ho.synthetic_vectors(n, d, seed=1234)
    q = ho.synthetic_queries(nq, d, seed=4321)
    ix = HipFlatIndex(d, metric)
    ix.add(x)
    assert ix.ntotal == n
    _check(ix, x, q, k, metric)
    st = ix.stats()
    assert st["queries"] == nq
    ix.close()


@pytest.mark.parametrize("metric", [ho.METRIC_IP, ho.METRIC_L2])
def test_unnormalised_and_scaled_vectors(gpu, metric):
    from hiprag import HipFlatIndex
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((3000, 256)) * rng.uniform(0.1, 30, size=(3000, 1))).astype(np.float32)
    q = (rng.standard_normal((9, 256)) * 7).astype(np.float32)
    ix = HipFlatIndex(256, metric)
    ix.add(x)
    s, i = ix.search(q, 10)
    es, ei = ho.flat_search(x, q, 10, metric)
    assert np.array_equal(i, ei)
    assert np.allclose(s, es, rtol=1e-6, atol=TOL)


@pytest.mark.parametrize("metric", [ho.METRIC_IP, ho.METRIC_L2])
def test_k_larger_than_ntotal_pads_with_minus_one(gpu, metric):
    from hiprag import HipFlatIndex
    x = ho.synthetic_vectors(5, 64, seed=1)
    ix = HipFlatIndex(64, metric)
    ix.add(x)
    s, i = _check(ix, x, ho.synthetic_queries(2, 64), 8, metric)
    assert np.all(i[:, 5:] == -1)


def test_empty_index_returns_padding(gpu):
    from hiprag import HipFlatIndex
    ix = HipFlatIndex(32, "l2")
    s, i = ix.search(np.zeros((2, 32), np.float32), 3)
    assert np.all(i == -1) and np.all(s == FLT_MAX)


@pytest.mark.parametrize("metric", [ho.METRIC_IP, ho.METRIC_L2])
def test_zero_query_is_all_ties_lowest_ids_win(gpu, metric):
    """embed_single('') returns the zero vector (rag/providers/hf/embeddings.py:47-48): IP ties everywhere."""
    from hiprag import HipFlatIndex
    x = ho.synthetic_vectors(3000, 128, seed=3)
    ix = HipFlatIndex(128, metric)
    ix.add(x)
    q = np.zeros((1, 128), np.float32)
    s, i = _check(ix, x, q, 10, metric)
    if metric == ho.METRIC_IP:
        assert list(i[0]) == list(range(10))
        st = ix.stats()                                 # the certificate must have refused the first round: all 188 groups tie,
        assert st["roundb_queries"] + st["fallback_queries"] >= 1   # round B re-scores them all


@pytest.mark.parametrize("metric", [ho.METRIC_IP, ho.METRIC_L2])
def test_duplicate_rows_tie_break_by_id(gpu, metric):
    from hiprag import HipFlatIndex
    base = ho.synthetic_vectors(40, 64, seed=9)
    x = np.concatenate([base] * 60, axis=0)          # every row appears 60 times, spread over many groups
    ix = HipFlatIndex(64, metric)
    ix.add(x)
    q = base[:4] + 0.01 * ho.synthetic_vectors(4, 64, seed=10)
    _check(ix, x, q, 25, metric)


def test_incremental_add_matches_single_add(gpu):
    from hiprag import HipFlatIndex
    x = ho.synthetic_vectors(1000, 96, seed=11)
    ix = HipFlatIndex(96, "ip")
    for lo, hi in [(0, 1), (1, 33), (33, 500), (500, 1000)]:
        ix.add(x[lo:hi])
    assert ix.ntotal == 1000
    for row in (0, 31, 32, 499, 999):
        assert np.array_equal(ix.reconstruct(row), x[row])
    _check(ix, x, ho.synthetic_queries(5, 96), 10, ho.METRIC_IP)


def test_id_base_offsets_ids(gpu):
    from hiprag import HipFlatIndex
    x = ho.synthetic_vectors(500, 64, seed=12)
    ix = HipFlatIndex(64, "ip")
    ix.add(x)
    ix.set_id_base(1_000_000)
    _check(ix, x, ho.synthetic_queries(3, 64), 5, ho.METRIC_IP, id_base=1_000_000)


def test_save_load_roundtrip(gpu, tmp_path):
    from hiprag import HipFlatIndex
    x = ho.synthetic_vectors(777, 200, seed=13)
    ix = HipFlatIndex(200, "l2")
    ix.add(x)
    p = str(tmp_path / "a.hipidx")
    ix.save(p)
    ix2 = HipFlatIndex.load(p)
    assert ix2.ntotal == 777 and ix2.d == 200 and ix2.metric == ho.METRIC_L2
    _check(ix2, x, ho.synthetic_queries(4, 200), 10, ho.METRIC_L2)


def test_more_than_32_queries_and_planted_neighbours(gpu):
    from hiprag import HipFlatIndex
    x = ho.synthetic_vectors(8192, 1024, seed=1234)
    rng = np.random.default_rng(77)
    rows = rng.integers(0, 8192, size=70)
    q = x[rows] + 0.05 * rng.standard_normal((70, 1024)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    ix = HipFlatIndex(1024, "ip")
    ix.add(x)
    s, i = _check(ix, x, q.astype(np.float32), 10, ho.METRIC_IP)
    assert np.array_equal(i[:, 0], rows)


def test_bad_arguments_raise(gpu):
    from hiprag import HipFlatIndex, HipRagError
    with pytest.raises(HipRagError):
        HipFlatIndex(0, "ip")
    with pytest.raises(HipRagError):
        HipFlatIndex(5000, "ip")           # beyond the LDS query tile
    ix = HipFlatIndex(8, "ip")
    with pytest.raises(ValueError):
        ix.add(np.zeros((2, 9), np.float32))
    with pytest.raises(HipRagError):
        ix.search(np.zeros((1, 8), np.float32), 0)


def test_device_api_and_merge_of_emulated_shards(gpu):
    """Row-sharded search == unsharded search bit for bit, shards emulated on one GPU."""
    import torch
    from hiprag import HipFlatIndex, merge_topk_device
    n, d, nq, k = 6000, 256, 12, 10
    x = ho.synthetic_vectors(n, d, seed=21)
    q = ho.synthetic_queries(nq, d, seed=22)
    bounds = [0, 1501, 1502, 4000, 6000]            # ragged, not multiples of 32
    for metric in (ho.METRIC_IP, ho.METRIC_L2):
        parts_s, parts_i = [], []
        qd = torch.from_numpy(q).cuda()
        shards = []
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            ix = HipFlatIndex(d, metric)
            ix.add(x[lo:hi])
            ix.set_id_base(lo)
            s64, s32, ids = ix.search_device(qd, k)
            parts_s.append(s64)
            parts_i.append(ids)
            shards.append(ix)
        m64, m32, mids = merge_topk_device(torch.stack(parts_s), torch.stack(parts_i), k, metric)
        torch.cuda.synchronize()
        full = HipFlatIndex(d, metric)
        full.add(x)
        fs, fi = full.search(q, k)
        assert np.array_equal(mids.cpu().numpy(), fi)
        assert np.array_equal(m32.cpu().numpy(), fs)          # bit for bit
        es, ei = ho.flat_search(x, q, k, metric)
        assert np.array_equal(fi, ei)


def test_two_stream_pipeline_matches_oracle(gpu):
    """ShardedFlatIndex (world 1) keeps one batch in flight over two streams / two workspace slots."""
    import torch
    from hiprag import HipFlatIndex
    from hiprag.sharded import ShardedFlatIndex
    n, d, k = 9000, 512, 10
    x = ho.synthetic_vectors(n, d, seed=31)
    q = ho.synthetic_queries(32 * 7 + 5, d, seed=32)
    ix = HipFlatIndex(d, "ip")
    ix.add(x)
    sh = ShardedFlatIndex(ix, 0)
    qd = torch.from_numpy(q).cuda()
    tickets, outs = None, []
    for o in range(0, q.shape[0], 32):
        t = sh.search_begin(qd[o:o + 32], k)
        if tickets is not None:
            outs.append(tuple(x.clone() for x in sh.search_end(tickets)))   # results live in slot-owned buffers
        tickets = t
    outs.append(tuple(x.clone() for x in sh.search_end(tickets)))
    torch.cuda.synchronize()
    ids = torch.cat([o[2] for o in outs]).cpu().numpy()
    s32 = torch.cat([o[1] for o in outs]).cpu().numpy()
    es, ei = ho.flat_search(x, q, k, ho.METRIC_IP)
    assert np.array_equal(ids, ei)
    assert np.allclose(s32, es, rtol=0, atol=TOL)
    s64b, s32b, idsb = sh.search_device(qd, k)
    torch.cuda.synchronize()
    assert np.array_equal(idsb.cpu().numpy(), ei)


@pytest.mark.parametrize("mode", ["q64", "bf16"])
@pytest.mark.parametrize("metric", [ho.METRIC_IP, ho.METRIC_L2])
def test_all_scan_operand_modes_give_the_same_exact_results(gpu, monkeypatch, mode, metric):
    """HIPRAG_SCAN_MODE picks what the scan streams (the bf16 filter copy / the fp32 rows split on the fly); the certificate
    + fp64 re-score make the RESULT identical in both modes, including ties and a zero query."""
    from hiprag import HipFlatIndex
    monkeypatch.setenv("HIPRAG_SCAN_MODE", mode)
    n, d, k = 7000, 512, 10
    x = ho.synthetic_vectors(n, d, seed=81)
    x[100:140] = x[5]                                    # 40 exact duplicates of row 5 spread over two blocks
    q = ho.synthetic_queries(70, d, seed=82)
    q[3] = x[5]
    q[4] = 0
    ix = HipFlatIndex(d, metric)
    assert ix.pass_queries == 64
    ix.add(x)
    _check(ix, x, q, k, metric)
    _check(ix, x, q[:1], 50, metric)
    st = ix.stats()                                      # the zero query ties all 438 groups: more than the finish re-scores,
    assert st["fallback_queries"] >= 1                   # so it took the exhaustive path


def test_unknown_scan_mode_is_rejected(gpu, monkeypatch):
    from hiprag import HipFlatIndex, HipRagError
    monkeypatch.setenv("HIPRAG_SCAN_MODE", "split")      # rounds 1-2 had four operand modes; two are left
    with pytest.raises(HipRagError):
        HipFlatIndex(64, ho.METRIC_IP)


@pytest.mark.parametrize("mode", ["q64", "bf16"])
@pytest.mark.parametrize("nq", [65, 129, 256, 1100])
def test_multi_pass_launches(gpu, monkeypatch, mode, nq):
    """One scan launch runs several passes back to back (cyclic piece stream, query tile re-staged per pass): ragged
    last passes, launches of exactly launch_queries and more than one launch, on an index small enough that some waves
    of the grid have no blocks at all, and on one where every wave has several."""
    from hiprag import HipFlatIndex
    monkeypatch.setenv("HIPRAG_SCAN_MODE", mode)
    for n, d in ((900, 96), (70001, 256)):
        x = ho.synthetic_vectors(n, d, seed=91)
        q = ho.synthetic_queries(nq, d, seed=92)
        ix = HipFlatIndex(d, ho.METRIC_IP)
        ix.add(x)
        assert ix.launch_queries == 1024                                  # small index: 16 passes per launch
        _check(ix, x, q, 10, ho.METRIC_IP)


def test_deep_k_on_the_candidate_lists_and_beyond(gpu):
    """k = 50 (the reference's retrieval depth, page_retriever.py:81) and k = 128 run on the candidate lists like k = 10 --
    same pass count, no exhaustive path; k = 129 is past what the finish holds and is answered by the exhaustive path alone
    (exact, slow).  70001 rows = 4376 groups: the in-scan filter is on."""
    from hiprag import HipFlatIndex
    n, d = 70001, 384
    x = ho.synthetic_vectors(n, d, seed=95)
    q = ho.synthetic_queries(100, d, seed=96)
    ix = HipFlatIndex(d, ho.METRIC_IP)
    ix.add(x)
    p0 = ix.stats()["passes"]
    for k in (128, 50, 10, 1):
        _check(ix, x, q, k, ho.METRIC_IP)
    assert ix.stats()["passes"] - p0 == 8                        # 100 queries = 2 passes of 64, four times
    assert ix.stats()["fallback_queries"] == 0
    _check(ix, x, q[:3], 129, ho.METRIC_IP)
    assert ix.stats()["fallback_queries"] == 3


def test_extension_of_the_rescored_prefix_ties_and_overflow(gpu):
    """Both scales: 120 exact copies of one row, each in a group of its own, tie for the top 50 of the query that equals
    them -- more tied groups than the first batch re-scores (k + k/2 = 75), fewer than the finish may re-score (192) -- so
    that query is settled by the EXTENSION step of the finish, lowest ids first.  A zero query ties EVERY group: the list
    overflows and the exhaustive path answers.  Results exact throughout."""
    from hiprag import HipFlatIndex
    n, d, k = 50000, 256, 50
    x = ho.synthetic_vectors(n, d, seed=97)
    copies = 1000 + 37 * np.arange(120)
    x[copies] = x[7]
    q = ho.synthetic_queries(70, d, seed=98)
    q[1] = x[7]
    q[2] = 0
    for metric in (ho.METRIC_L2, ho.METRIC_IP):
        ix = HipFlatIndex(d, metric)
        ix.add(x)
        s, i = _check(ix, x, q, k, metric)
        assert list(i[1]) == [7] + list(copies[:49])
        st = ix.stats()
        assert st["roundb_queries"] >= 1 and 1 <= st["fallback_queries"] <= 2


@pytest.mark.parametrize("mode", ["bf16", "q64"])
def test_full_size_1m_x_1024_properties(gpu, monkeypatch, mode):
    """Both scan modes a number is quoted for (a wrong wait in the scan's load ring produces wrong scan values the
    certificate cannot see, and only this full-size test catches them).
    BASELINE configs[1] at full size (1M x 1024, top-10), where the CPU oracle would take minutes: size-independent
    properties instead -- (a) 256 planted queries (row + 5 % noise) find their row first, (b) every result list is sorted by
    (score desc, id asc) and repeats bit for bit, (c) eight row shards merged == the unsharded index bit for bit,
    (d) the top-10 of 8 queries equals an independent fp64 ranking computed chunk by chunk with torch on the GPU."""
    import torch
    from hiprag import HipFlatIndex, merge_topk_device
    monkeypatch.setenv("HIPRAG_SCAN_MODE", mode)
    N, d, k, chunk = 1_000_000, 1024, 10, 125_000
    dev = torch.device("cuda", 0)

    def rows_of(c):
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + c)
        x = torch.randn((chunk, d), generator=g, device=dev, dtype=torch.float32)
        return x / x.norm(dim=1, keepdim=True)

    full = HipFlatIndex(d, ho.METRIC_IP)
    shards = []
    for c in range(N // chunk):
        x = rows_of(c)
        full.add_device(x)
        sh = HipFlatIndex(d, ho.METRIC_IP)
        sh.add_device(x)
        sh.set_id_base(c * chunk)
        shards.append(sh)
    assert full.ntotal == N
    # (a) planted neighbours
    planted = torch.arange(256, device=dev) * 3907 + 11
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    src = torch.cat([rows_of(int(p) // chunk)[int(p) % chunk][None] for p in planted.tolist()])
    noise = torch.randn(src.shape, generator=g, device=dev)
    q = src + 0.05 * noise / noise.norm(dim=1, keepdim=True)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    s64, s32, ids = full.search_device(q, k)
    torch.cuda.synchronize()
    assert torch.equal(ids[:, 0], planted)
    # (b) canonical order, determinism
    sc, idn = s64.cpu().numpy(), ids.cpu().numpy()
    assert np.all((sc[:, :-1] > sc[:, 1:]) | ((sc[:, :-1] == sc[:, 1:]) & (idn[:, :-1] < idn[:, 1:])))
    s64b, _, idsb = full.search_device(q, k)
    torch.cuda.synchronize()
    assert torch.equal(idsb, ids) and torch.equal(s64b, s64)
    # (c) 8 row shards + canonical merge == unsharded, bit for bit
    ps, pi = [], []
    for sh in shards:
        a, _, b = sh.search_device(q, k)
        ps.append(a.clone())
        pi.append(b.clone())
    m64, _, mids = merge_topk_device(torch.stack(ps), torch.stack(pi), k, ho.METRIC_IP)
    torch.cuda.synchronize()
    assert torch.equal(mids, ids) and torch.equal(m64, s64)
    # (d) independent fp64 ranking of 8 queries
    q8 = q[:8].double()
    best_s = torch.full((8, 0), 0.0, dtype=torch.float64, device=dev)
    best_i = torch.zeros((8, 0), dtype=torch.int64, device=dev)
    for c in range(N // chunk):
        sc_c = q8 @ rows_of(c).double().T                                   # [8, chunk] fp64
        ts, ti = torch.topk(sc_c, k, dim=1)
        best_s = torch.cat([best_s, ts], 1)
        best_i = torch.cat([best_i, ti + c * chunk], 1)
    order = torch.argsort(best_s, dim=1, descending=True, stable=True)[:, :k]
    ref_i = torch.gather(best_i, 1, order)
    ref_s = torch.gather(best_s, 1, order)
    assert torch.equal(ref_i, ids[:8])
    assert torch.allclose(ref_s, s64[:8], rtol=0, atol=1e-12)
    assert full.stats()["fallback_queries"] == 0


def test_spare_cus_change_the_partition_not_the_results(gpu):
    """hipidx_set_spare_cus re-partitions the scan (fewer workgroups, other block ranges per wave, other group slots);
    results stay exact, also for a launch in which some waves have no blocks at all."""
    from hiprag import HipFlatIndex
    n, d = 40000, 512
    x = ho.synthetic_vectors(n, d, seed=61)
    q = ho.synthetic_queries(130, d, seed=62)
    ix = HipFlatIndex(d, ho.METRIC_IP)
    ix.add(x)
    for spare in (0, 8, 200, 0):
        ix.set_spare_cus(spare)
        _check(ix, x, q, 10, ho.METRIC_IP)
    with pytest.raises(Exception):
        ix.set_spare_cus(100000)


def test_randomised_shapes_against_the_oracle(gpu, monkeypatch):
    """Forty seeded random configurations -- rows, width, k, batch size, metric, operand mode, spare CUs, duplicated and
    zero rows, un-normalised data -- through search(), every one compared id for id with the fp64 oracle."""
    from hiprag import HipFlatIndex
    rng = np.random.default_rng(20260101)
    for case in range(40):
        n = int(rng.choice([1, 31, 33, 500, 4097, 9000, 30011]))
        d = int(rng.choice([8, 100, 128, 384, 1000, 1024]))
        k = int(rng.choice([1, 5, 10, 31, 50, 57, 64, 128]))
        nq = int(rng.choice([1, 2, 63, 64, 65, 200, 300]))
        metric = [ho.METRIC_IP, ho.METRIC_L2][case % 2]
        mode = ["bf16", "bf16", "q64", "q64", "bf16"][int(rng.integers(5))]
        monkeypatch.setenv("HIPRAG_SCAN_MODE", mode)
        x = ho.synthetic_vectors(n, d, seed=1000 + case)
        q = ho.synthetic_queries(nq, d, seed=2000 + case)
        if case % 3 == 0:
            x *= rng.uniform(0.1, 30.0, size=(n, 1)).astype(np.float32)       # un-normalised rows
        if case % 4 == 1 and n > 40:
            x[rng.integers(0, n, size=20)] = x[3]                              # duplicates -> exact ties
        if case % 5 == 2:
            x[n // 2] = 0
            q[0] = 0                                                           # zero row / zero query
        ix = HipFlatIndex(d, metric)
        ix.add(x[: n // 2])
        ix.add(x[n // 2:])
        if case % 6 == 3:
            ix.set_spare_cus(int(rng.choice([1, 8, 100])))
        try:
            _check(ix, x, q, k, metric)
        except AssertionError as e:
            raise AssertionError(f"case {case}: n={n} d={d} k={k} nq={nq} metric={metric} mode={mode}: {e}")


@pytest.mark.parametrize("tails", ["in_stream_order", "aside", "aside_one_tail_stream"])
def test_deep_pipeline_with_changing_batch_shapes(gpu, tails, monkeypatch):
    """Six launches in flight over the eight workspace slots, batch size and k changing from launch to launch (every slot's
    buffers are re-shaped while older launches are still running), in both arrangements of the finish: on the slot streams
    beside later scans, which run on the library's high-priority scan stream (the default), and behind its scan on the
    caller's stream (the A/B arrangement).  The number of CUs the scan leaves free changes between launches in flight too
    (hipidx_set_spare_cus takes effect per launch, no synchronisation)."""
    import torch
    from collections import deque
    from hiprag import HipFlatIndex
    from hiprag.sharded import ShardedFlatIndex
    n, d = 20000, 256
    x = ho.synthetic_vectors(n, d, seed=71)
    q = ho.synthetic_queries(256, d, seed=72)
    ix = HipFlatIndex(d, "ip")
    ix.add(x)
    if tails == "aside_one_tail_stream":
        monkeypatch.setenv("HIPRAG_SIDE_STREAMS", "1")         # every slot's tails on ONE stream (default: two, taking turns)
    sh = ShardedFlatIndex(ix, 0, tails_aside=tails.startswith("aside"))
    qd = torch.from_numpy(q).cuda()
    rng = np.random.default_rng(5)
    plan = [(int(rng.choice([1, 17, 64, 65, 200, 256])), int(rng.choice([1, 10, 50]))) for _ in range(40)]
    pending, got = deque(), []
    for i, (nq, k) in enumerate(plan):
        ix.set_spare_cus([48, 0, 200, 17][i % 4])
        assert ix.spare_cus == [48, 0, 200, 17][i % 4]
        pending.append((nq, k, sh.search_begin(qd[:nq], k)))
        if len(pending) >= 6:
            a, b, t = pending.popleft()
            got.append((a, b, tuple(v.clone() for v in sh.search_end(t))))
    while pending:
        a, b, t = pending.popleft()
        got.append((a, b, tuple(v.clone() for v in sh.search_end(t))))
    torch.cuda.synchronize()
    expect = {}
    for nq, k, (s64, s32, ids) in got:
        if (nq, k) not in expect:
            expect[(nq, k)] = ho.flat_search(x, q[:nq], k, ho.METRIC_IP)
        es, ei = expect[(nq, k)]
        assert np.array_equal(ids.cpu().numpy(), ei), (nq, k)
        assert np.allclose(s32.cpu().numpy(), es, rtol=0, atol=TOL)


@pytest.mark.parametrize("metric", [ho.METRIC_IP, ho.METRIC_L2])
def test_batches_of_several_launches_are_pipelined_inside_the_library(gpu, monkeypatch, metric):
    """hipidx_search / hipidx_search_dev with more queries than one launch takes: the library chains the scans on its scan
    stream and runs every finish but the last beside the next scan (start gate), results ordered on the caller's stream.
    Launch size forced down to 64 queries so that 300 queries are five launches and 700 eleven (more than the eight
    workspace slots: slots are reused inside one call); a caller's own stream; k = 10 and 50; ids equal the oracle's and the
    call leaves the index's spare-CU setting alone."""
    import torch
    from hiprag import HipFlatIndex
    monkeypatch.setenv("HIPRAG_LAUNCH_QUERIES", "64")
    n, d = 30011, 256
    x = ho.synthetic_vectors(n, d, seed=95)
    q = ho.synthetic_queries(700, d, seed=96)
    ix = HipFlatIndex(d, metric)
    ix.add(x)
    assert ix.launch_queries == 64
    ix.set_spare_cus(5)
    for nq, k in ((300, 10), (700, 50), (65, 1)):
        _check(ix, x, q[:nq], k, metric)                                   # host arrays in and out (null stream)
        _, ei = ho.flat_search(x, q[:nq], k, metric)
        mine = torch.cuda.Stream()
        with torch.cuda.stream(mine):
            qd = torch.from_numpy(q[:nq]).cuda() * 1.0                     # produced on the caller's stream right before
            out = ix.search_device(qd, k)
            back = out[2].cpu()                                            # consumed on it right after
        mine.synchronize()
        assert np.array_equal(back.numpy(), ei), (nq, k)
    assert ix.spare_cus == 5


def test_start_gate_orders_a_finish_behind_the_start_of_the_next_scan(gpu):
    """hipidx_gate_tail_dev by hand, the way a C host would use it: scans chained on the library's scan stream, the finish of
    step i on a tail stream behind (a) the end of scan i and (b) the START of scan i + 1.  The library refuses a gate that
    nothing would ever open (no scan launched behind the slot's yet) -- a stream waiting for it would hang -- and the gated
    pipeline returns the oracle's lists."""
    import ctypes
    import torch
    from hiprag import HipFlatIndex, HipRagError
    from hiprag import _native as nat
    n, d, k = 30000, 256, 10
    x = ho.synthetic_vectors(n, d, seed=91)
    q = ho.synthetic_queries(192, d, seed=92)
    ix = HipFlatIndex(d, "ip")
    ix.add(x)
    ix.set_spare_cus(48)
    ptr = ctypes.c_void_p()
    nat.call("hiprag_scan_stream", 0, ctypes.byref(ptr))
    scan = torch.cuda.ExternalStream(ptr.value)
    tail = torch.cuda.Stream()
    qd = torch.from_numpy(q).cuda()
    torch.cuda.synchronize()
    outs, scanned = [], []
    for i in range(3):
        qi = qd[64 * i:64 * (i + 1)]
        ix.search_begin(qi, k, slot=i, stream=scan.cuda_stream)
        ev = torch.cuda.Event()
        ev.record(scan)
        scanned.append(ev)
        if i == 0:
            with pytest.raises(HipRagError, match="no scan has been launched after"):
                ix.gate_tail(0, tail.cuda_stream)
        else:                                   # the tails of step i - 1, gated on the scan just launched
            tail.wait_event(scanned[i - 1])
            ix.gate_tail(i - 1, tail.cuda_stream)
            out = (torch.empty((64, k), dtype=torch.float64, device="cuda"), torch.empty((64, k), dtype=torch.float32, device="cuda"),
                   torch.empty((64, k), dtype=torch.int64, device="cuda"))
            ix.search_finish(qd[64 * (i - 1):64 * i], k, i - 1, out, stream=tail.cuda_stream)
            outs.append(out)
    tail.wait_event(scanned[2])                 # the last step: no scan behind it, no gate
    out = tuple(torch.empty((64, k), dtype=t, device="cuda") for t in (torch.float64, torch.float32, torch.int64))
    ix.search_finish(qd[128:192], k, 2, out, stream=tail.cuda_stream)
    outs.append(out)
    torch.cuda.synchronize()
    _, ei = ho.flat_search(x, q, k, ho.METRIC_IP)
    got = np.concatenate([o[2].cpu().numpy() for o in outs])
    assert np.array_equal(got, ei)


@pytest.mark.parametrize("mode", ["bf16", "q64"])
def test_near_ties_below_the_scan_resolution_stay_exact(gpu, monkeypatch, mode):
    """Adversarial for the certificate: 3000 rows that differ from each other by 1e-6 .. 1e-4 relative -- far below what
    the bf16 operands of the scan resolve -- surround the k-th score, together with random rows.  The scan cannot order
    them; the certificate has to notice (round B or the exhaustive path) and the fp64 re-score decides: ids must equal
    the oracle's, for unit and for scaled queries."""
    from hiprag import HipFlatIndex
    monkeypatch.setenv("HIPRAG_SCAN_MODE", mode)
    rng = np.random.default_rng(123)
    n, d, k = 20000, 512, 20
    x = ho.synthetic_vectors(n, d, seed=55)
    u = x[11].copy()
    close = rng.choice(np.arange(100, n), size=3000, replace=False)
    pert = rng.standard_normal((3000, d)).astype(np.float32)
    pert /= np.linalg.norm(pert, axis=1, keepdims=True)
    scale = (10.0 ** rng.uniform(-6, -4, size=(3000, 1))).astype(np.float32)
    x[close] = u[None, :] + scale * pert
    q = np.stack([u, 3.7 * u, u + 1e-5 * pert[0], -u]).astype(np.float32)
    for metric in (ho.METRIC_IP, ho.METRIC_L2):
        ix = HipFlatIndex(d, metric)
        ix.add(x)
        _check(ix, x, q, k, metric)
        st = ix.stats()
        assert st["roundb_queries"] + st["fallback_queries"] >= 1


@pytest.mark.parametrize("mode", ["bf16", "q64"])
def test_certificate_with_bf16_exact_inputs_and_ulp_level_ties(gpu, monkeypatch, mode):
    """VERDICT r1 (certificate coverage hole).  With bf16-representable rows AND queries both truncation terms of the
    certificate vanish (dq2 = dx2 = 0: integer-valued or pre-quantised embeddings) and eps collapses to the fp32
    accumulation bound of the scan alone, (d_pad + 80) * 2^-24 * |q||x| -- an assumption about the MFMA's internal rounding
    that no other test reaches.  Here 3000 rows at d = 1024 equal one base row except for the signs of 64 coordinates where
    the query holds tiny (bf16-exact) values, so their exact scores differ by 1e-9 .. 4e-6 relative: from far below one fp32 ulp of
    the score to a few dozen ulps, all inside what fp32 accumulation can blur.  Every scan mode must return the fp64
    order (ids equal to the oracle's), for a unit-scale and a x4 query, both metrics."""
    import torch
    from hiprag import HipFlatIndex
    monkeypatch.setenv("HIPRAG_SCAN_MODE", mode)
    rng = np.random.default_rng(2026)
    n, d, k = 12000, 1024, 20

    def bf16_exact(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()

    x = bf16_exact(rng.standard_normal((n, d)) / 32.0)
    u = x[11].copy()
    close = rng.choice(np.arange(100, n), size=3000, replace=False)
    x[close] = u[None, :]
    # ... differing from u only where the query is tiny, and only by SIGNS: every variant has the same norm, so the rows
    # are near-ties under L2 as well as under inner product
    x[close, :64] = bf16_exact(np.abs(rng.standard_normal(64)) / 32.0)[None, :] * rng.choice([-1.0, 1.0], size=(3000, 64))
    x[close[:40]] = x[close[0]]                                              # and 40 exact duplicates: ties by id
    q0 = u.copy()
    q0[:64] = bf16_exact(rng.standard_normal(64) * 2.0 ** -17)
    # a second, smaller family (150 variants of row 13): few enough groups within eps for round B to settle it, where the
    # 3000-row family overflows round B and takes the exhaustive path
    few = np.setdiff1d(np.arange(100, n), close)[::55][:150]             # spread out: (nearly) one per 16-row group
    x[few] = x[13][None, :]
    x[few, :64] = np.abs(x[13, :64])[None, :] * rng.choice([-1.0, 1.0], size=(150, 64))
    q1 = x[13].copy()
    q1[:64] = q0[:64]
    q = np.stack([q0, 4.0 * q0, q1, -q0]).astype(np.float32)
    assert np.array_equal(bf16_exact(x), x) and np.array_equal(bf16_exact(q), q)
    s64 = ho.all_scores_f64(x[close], q0)
    gaps = np.diff(np.sort(s64))
    assert np.median(gaps[gaps > 0]) < 6e-8 * abs(s64).max()                 # typical neighbour gap below one fp32 ulp
    for metric in (ho.METRIC_IP, ho.METRIC_L2):
        ix = HipFlatIndex(d, metric)
        ix.add(x)
        _check(ix, x, q, k, metric)
        _check(ix, x, q[:1], 50, metric)
        st = ix.stats()
        assert st["fallback_queries"] >= 1           # the scan alone cannot have ordered the large family ...
        assert st["roundb_queries"] >= 1             # ... nor the small one, which round B settles
