"""
Multi-rank path.  CPU (gloo, world_size 2): the exchange protocol -- shard assignment, the packed [2,nq,k] int64
layout, the single all-gather and the canonical merge -- with the oracle standing in for the per-rank GPU search.
GPU (-m gpu): the real thing, two ranks sharing cuda:0 over gloo (RCCL refuses two ranks on one device), asserting
sharded == unsharded == oracle bit for bit through ShardedFlatIndex's two-stream pipeline.
"""
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def list_gather(pack, gathered, group=None, async_op=True):
    """The exchange over a backend without all_gather_into_tensor (gloo: the CPU protocol test and the two-ranks-on-one-GPU
    rehearsals) -- same memory, list form.  The product's gather is RCCL's all_gather_into_tensor (hiprag/sharded.py)."""
    import torch.distributed as dist
    return dist.all_gather([gathered[r] for r in range(gathered.shape[0])], pack, group=group, async_op=async_op)


def _setup(rank, world, port):
    for p in (REPO, os.path.join(REPO, "intool-rag_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def _cpu_worker(rank, world, port, q_out):
    import torch
    dist = _setup(rank, world, port)
    from hiprag.sharded import chunks_of_rank, shard_bounds
    from oracle import hybrid_oracle as ho
    try:
        n, d, nq, k = 4001, 48, 6, 10
        x = ho.synthetic_vectors(n, d, seed=51)
        q = ho.synthetic_queries(nq, d, seed=52)
        lo, hi = shard_bounds(n, world)[rank]
        for metric in (ho.METRIC_IP, ho.METRIC_L2):
            _, ids, s64 = ho.flat_search(x[lo:hi], q, k, metric, id_base=lo, return_f64=True)
            pack = torch.empty((2, nq, k), dtype=torch.int64)
            pack[0] = torch.from_numpy(s64.view(np.int64))
            pack[1] = torch.from_numpy(ids)
            gathered = torch.empty((world, 2, nq, k), dtype=torch.int64)
            list_gather(pack, gathered, None, async_op=True).wait()
            parts_s = [gathered[r, 0].numpy().view(np.float64) for r in range(world)]
            parts_i = [gathered[r, 1].numpy() for r in range(world)]
            ms, mi = ho.merge_partial_topk(parts_s, parts_i, k, metric)
            fs, fi, f64 = ho.flat_search(x, q, k, metric, return_f64=True)
            assert np.array_equal(mi, fi) and np.array_equal(ms, f64)
        # hybrid: dense + BM25 partial lists of the same document range in ONE packed all-gather
        # ([leg, {score bits, ids}, nq, depth]), each leg merged globally, RRF after the merge
        depth = 20
        p = ho.synthetic_postings(n, n_terms=256, seed=53)
        sq = ho.synthetic_sparse_queries(nq, n_terms=256, terms_per_query=4, seed=54, min_rank=2)
        _, dids, d64 = ho.flat_search(x[lo:hi], q, depth, ho.METRIC_IP, id_base=lo, return_f64=True)
        from hiprag import PostingsCSR
        sh = PostingsCSR(p.n_docs, p.n_terms, p.offsets, p.doc_ids, p.impacts).shard(lo, hi)   # local ids, global idf/avgdl
        bs, bids = ho.bm25_search(ho.Postings(sh.n_docs, sh.n_terms, sh.offsets, sh.doc_ids, sh.impacts), sq, depth)
        bids = np.where(bids >= 0, bids + lo, bids)
        pack = torch.empty((2, 2, nq, depth), dtype=torch.int64)
        pack[0, 0] = torch.from_numpy(d64.view(np.int64))
        pack[0, 1] = torch.from_numpy(dids)
        pack[1, 0] = torch.from_numpy(bs.astype(np.float64).view(np.int64))
        pack[1, 1] = torch.from_numpy(bids)
        gathered = torch.empty((world,) + tuple(pack.shape), dtype=torch.int64)
        list_gather(pack, gathered, None, async_op=True).wait()
        g = gathered.numpy()
        _, gd = ho.merge_partial_topk([g[r, 0, 0].view(np.float64) for r in range(world)], [g[r, 0, 1] for r in range(world)],
                                      depth, ho.METRIC_IP)
        _, gb = ho.merge_partial_topk([g[r, 1, 0].view(np.float64) for r in range(world)], [g[r, 1, 1] for r in range(world)],
                                      depth, ho.METRIC_IP)
        fs_, fi_ = ho.rrf_fuse(gd, gb, k)
        _, di_full = ho.flat_search(x, q, depth, ho.METRIC_IP)
        _, bi_full = ho.bm25_search(p, sq, depth)
        es_, ei_ = ho.rrf_fuse(di_full, bi_full, k)
        assert np.array_equal(fi_, ei_) and np.array_equal(fs_, es_)
        # ranks whose shards differ by a block may derive different launch sizes: the agreed batch size is the minimum,
        # and the debug check catches ranks entering a collective with different batch shapes
        from hiprag.sharded import agree_min, check_same_shape
        assert agree_min(448 + 64 * rank, 0) == 448
        check_same_shape((nq, k))
        try:
            check_same_shape((nq + rank, k))
            raise AssertionError("shape mismatch not detected")
        except RuntimeError:
            pass
        owned = [chunks_of_rank(32, world, r) for r in range(world)]
        assert sorted(sum(owned, [])) == list(range(32)) and all(o == list(range(o[0], o[-1] + 1)) for o in owned)
        q_out.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        q_out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def _run(worker, world=2, timeout=180):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q_out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q_out.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, "ok") for r in range(world)], results


def test_exchange_protocol_world2_gloo_cpu():
    _run(_cpu_worker, world=2)


def test_shard_bounds_cover_and_balance():
    sys.path.insert(0, os.path.join(REPO, "intool-rag_amd"))
    from hiprag.sharded import shard_bounds
    for n in (0, 1, 7, 1_000_000, 10_000_001):
        for w in (1, 2, 3, 8):
            b = shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _gpu_worker(rank, world, port, q_out):
    import torch
    dist = _setup(rank, world, port)
    from hiprag import HipFlatIndex
    from hiprag.sharded import ShardedFlatIndex, shard_bounds
    from oracle import hybrid_oracle as ho
    try:
        torch.cuda.set_device(0)
        n, d, k = 7001, 256, 10
        x = ho.synthetic_vectors(n, d, seed=61)
        q = ho.synthetic_queries(32 * 3 + 7, d, seed=62)
        lo, hi = shard_bounds(n, world)[rank]
        for metric in ("ip", "l2"):
            local = HipFlatIndex(d, metric)
            local.add(x[lo:hi])
            sh = ShardedFlatIndex(local, lo, gather=list_gather)
            assert sh.world == world
            s64, s32, ids = sh.search_device(torch.from_numpy(q).cuda(), k)
            torch.cuda.synchronize()
            es, ei = ho.flat_search(x, q, k, ho.METRIC_IP if metric == "ip" else ho.METRIC_L2)
            assert np.array_equal(ids.cpu().numpy(), ei)
            assert np.array_equal(s32.cpu().numpy(), es)
        q_out.put((rank, "ok"))
    except Exception as e:
        q_out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_index_two_ranks_on_one_gpu(gpu):
    _run(_gpu_worker, world=2, timeout=300)


def _gpu_hybrid_worker(rank, world, port, q_out):
    """ShardedHybrid (SURVEY 8e / configs[3] shape) with two ranks on one GPU over gloo: dense rows and postings split by
    the same document ranges, pipelined batches (the launch size is forced down to 64 so that 150 queries run as three
    pieces with several slots in flight), one packed all-gather per piece, fused lists == the unsharded oracle's."""
    import torch
    os.environ["HIPRAG_LAUNCH_QUERIES"] = "64"
    dist = _setup(rank, world, port)
    from hiprag import HipBM25, HipFlatIndex, PostingsCSR
    from hiprag.sharded import ShardedHybrid, shard_bounds
    from oracle import hybrid_oracle as ho
    try:
        torch.cuda.set_device(0)
        n, d, depth, k, nq = 9001, 128, 50, 10, 150
        x = ho.synthetic_vectors(n, d, seed=81)
        q = ho.synthetic_queries(nq, d, seed=82)
        p = ho.synthetic_postings(n, n_terms=512, seed=83)
        sq = ho.synthetic_sparse_queries(nq, n_terms=512, terms_per_query=5, seed=84, min_rank=4)
        lo, hi = shard_bounds(n, world)[rank]
        local = HipFlatIndex(d, "ip")
        local.add(x[lo:hi])
        full = PostingsCSR(p.n_docs, p.n_terms, p.offsets, p.doc_ids, p.impacts)
        sh = ShardedHybrid(local, HipBM25(full.shard(lo, hi)), lo, gather=list_gather)
        assert sh.world == world and sh.max_pass == 64, (sh.world, sh.max_pass)
        qd = torch.from_numpy(q).cuda()
        _, di = ho.flat_search(x, q, depth, ho.METRIC_IP)
        _, bi = ho.bm25_search(p, sq, depth)
        for (c, wd, ws) in [(60.0, 1.0, 1.0), (60.0, 0.7, 0.3)]:
            fs, fi = sh.search_device(qd, sq, depth=depth, k=k, c=c, w_dense=wd, w_sparse=ws)
            torch.cuda.synchronize()
            es, ei = ho.rrf_fuse(di, bi, k, c=c, w_a=wd, w_b=ws)
            gi, gs = fi.cpu().numpy(), fs.cpu().numpy()
            bad = np.flatnonzero((gi != ei).any(axis=1) | (gs != es).any(axis=1))
            if bad.size:    # say whether a repeat of the same call agrees (a race) or not (a wrong result)
                fs2, fi2 = sh.search_device(qd, sq, depth=depth, k=k, c=c, w_dense=wd, w_sparse=ws)
                torch.cuda.synchronize()
                again = bool(np.array_equal(fi2.cpu().numpy(), ei) and np.array_equal(fs2.cpu().numpy(), es))
                raise AssertionError(f"fused lists differ from the oracle's for queries {bad.tolist()[:24]} ({bad.size} of {nq}; "
                                     f"weights {wd}, {ws}); the same call repeated {'agrees' if again else 'differs again'}")
        q_out.put((rank, "ok"))
    except Exception as e:
        q_out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_hybrid_two_ranks_on_one_gpu(gpu):
    _run(_gpu_hybrid_worker, world=2, timeout=300)


@pytest.mark.gpu
def test_bench_rehearsal_two_ranks_under_torch_distributed_run(gpu):
    """bench.py --gpus 2 exactly as the driver launches it (python -m torch.distributed.run, one process per rank), except
    that both ranks share cuda:0 over gloo (RCCL refuses two ranks on one device): argument, seed, shard and JSON paths of
    the multi-GPU run are exercised on the one-GPU box."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--rows", "128000", "--backend", "gloo", "--share-gpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["world"] == 2 and d["config"]["rows_local"] == [64000, 64000]
    assert d["config"]["backend"] == "gloo" and d["config"]["allgather_payload_bytes_per_rank"] > 0
    assert d["value"] > 0 and d["fallback_queries"] == 0 and "legs" not in d and "cpu_baseline" not in d


def _bench_line(args, timeout=900):
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       env=env, cwd=REPO)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_when_no_launcher_is_around_it(gpu):
    """`python bench.py --gpus 2` as a plain command (what the driver ran at N = 1 in round 2): the parent starts the two
    ranks itself before touching the GPU and relays rank 0's line.  Rehearsal form: both ranks share cuda:0 over gloo."""
    r, d = _bench_line(["--gpus", "2", "--steps", "4", "--warmup", "2", "--rows", "128000", "--backend", "gloo", "--share-gpu"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["config"]["rows_local"] == [64000, 64000]
    assert [p["rank"] for p in d["per_rank"]] == [0, 1] and all(p["scan_ms"] > 0 for p in d["per_rank"])
    assert d["value"] > 0 and d["fallback_queries"] == 0 and "legs" not in d


@pytest.mark.gpu
def test_bench_two_rccl_ranks_on_a_one_gpu_box_fail_loudly(gpu):
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    r, d = _bench_line(["--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "128000"])
    assert r.returncode != 0 and d is None and "FATAL" in r.stderr


@pytest.mark.gpu
def test_bench_config3_workload_two_ranks_rehearsal(gpu):
    """--workload config3 (rows AND postings sharded by the same document ranges, hybrid top-10 through ShardedHybrid) at a
    rehearsal size: two ranks sharing cuda:0 over gloo, started by bench.py itself; every rank must end with the same list."""
    r, d = _bench_line(["--gpus", "2", "--workload", "config3", "--rows", "64000", "--steps", "3", "--warmup", "1",
                        "--backend", "gloo", "--share-gpu"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["config"]["rows_local"] == [32000, 32000]
    assert d["fused_lists_equal_on_all_ranks"] and d["value"] > 0 and d["config"]["allgather_payload_bytes_per_rank"] > 0
    assert all(p > 0 for p in d["config"]["postings_local"])


@pytest.mark.gpu
def test_bench_single_rank_over_rccl_runs_the_exchange_path(gpu):
    """The calls of the N > 1 run that a one-GPU box CAN execute on RCCL itself: bench.py --force-dist initialises the
    `nccl` process group with one rank (device_id form), agrees the launch size by all-reduce, and runs the per-step
    all_gather_into_tensor (async, on the slot streams) + merge of ShardedFlatIndex, the barriers, the MAX all-reduce of the
    elapsed time and all_gather_object -- everything but a second GPU on the wire.  Results must still be the exact ones
    (bench.py compares sampled ids with the CPU port when the baseline is on; here: no fallbacks, sane line)."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--force-dist", "--backend", "nccl", "--steps", "6",
           "--warmup", "2", "--rows", "256000", "--legs", "none", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["backend"] == "nccl" and d["config"]["world"] == 1 and d["config"]["rows_local"] == [256000]
    assert d["config"]["allgather_payload_bytes_per_rank"] > 0 and d["value"] > 0 and d["fallback_queries"] == 0


def _rccl_one_rank_worker(rank, world, port, q_out):
    """ShardedFlatIndex and ShardedHybrid with the exchange forced on a one-rank RCCL group: sharded == oracle."""
    import torch
    for p in (REPO, os.path.join(REPO, "intool-rag_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HIPRAG_FORCE_EXCHANGE="1",
                      HIPRAG_CHECK_SHAPES="1", HSA_ENABLE_IPC_MODE_LEGACY="0", HIPRAG_LAUNCH_QUERIES="64")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from hiprag import HipBM25, HipFlatIndex, PostingsCSR
    from hiprag.sharded import ShardedFlatIndex, ShardedHybrid
    from oracle import hybrid_oracle as ho
    try:
        n, d, depth, k, nq = 9001, 128, 50, 10, 150
        x = ho.synthetic_vectors(n, d, seed=91)
        q = ho.synthetic_queries(nq, d, seed=92)
        qd = torch.from_numpy(q).cuda()
        local = HipFlatIndex(d, "ip")
        local.add(x)
        sh = ShardedFlatIndex(local, 0)
        assert sh.exchange and sh.world == 1 and sh.max_pass == 64
        s64, s32, ids = sh.search_device(qd, k)          # three pipelined pieces, one all-gather each
        torch.cuda.synchronize()
        es, ei = ho.flat_search(x, q, k, ho.METRIC_IP)
        assert np.array_equal(ids.cpu().numpy(), ei) and np.array_equal(s32.cpu().numpy(), es)
        p = ho.synthetic_postings(n, n_terms=512, seed=93)
        sq = ho.synthetic_sparse_queries(nq, n_terms=512, terms_per_query=5, seed=94, min_rank=4)
        hy = ShardedHybrid(local, HipBM25(PostingsCSR(p.n_docs, p.n_terms, p.offsets, p.doc_ids, p.impacts)), 0)
        assert hy.exchange
        fs, fi = hy.search_device(qd, sq, depth=depth, k=k)
        torch.cuda.synchronize()
        _, di = ho.flat_search(x, q, depth, ho.METRIC_IP)
        _, bi = ho.bm25_search(p, sq, depth)
        efs, efi = ho.rrf_fuse(di, bi, k)
        assert np.array_equal(fi.cpu().numpy(), efi) and np.array_equal(fs.cpu().numpy(), efs)
        q_out.put((rank, "ok"))
    except Exception as e:
        import traceback
        q_out.put((rank, repr(e) + traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_classes_on_a_one_rank_rccl_group(gpu):
    _run(_rccl_one_rank_worker, world=1, timeout=300)
