"""CPU-side checks of bench.py's launcher logic and of the shard-local postings build (no GPU needed)."""
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "intool-rag_amd"))


def _bench(args, env_extra=None, drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    env = dict(os.environ)
    for k in drop:
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, timeout=300,
                          env=env, cwd=REPO)


def test_bench_refuses_more_ranks_than_gpus_instead_of_measuring_one():
    """`python bench.py --gpus 2` without a launcher starts its own ranks -- and on a box with fewer GPUs than ranks it must
    fail loudly (it used to print a note and measure one GPU).  This container has no GPU at all."""
    import torch
    if torch.cuda.device_count() >= 2:
        return   # a multi-GPU box: the launcher would really start two ranks; covered by the gpu-marked rehearsals
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "FATAL" in r.stderr and "GPU(s) are visible" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_bench_world_size_mismatch_is_fatal():
    """A launcher that starts another number of ranks than --gpus says: exit 2 before anything is measured."""
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"],
               env_extra={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "1"}, drop=())
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "WORLD_SIZE=3" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_shard_local_postings_build_equals_shard_of_the_full_build():
    """bench.py --workload config3 builds every rank's postings from its own documents only, with the collection-wide
    constants (df, N, avgdl) passed in: the result must equal PostingsCSR.shard() of the unsharded build bit for bit."""
    from hiprag import build_postings
    rng = np.random.default_rng(5)
    n, V = 3000, 257
    doc_len = 5 + (np.arange(n) * 2654435761) % 23
    doc = np.repeat(np.arange(n), doc_len)
    term = np.minimum((rng.pareto(1.1, doc.size) * 3).astype(np.int64), V - 1)
    full = build_postings(doc, term, n, V, doc_len)
    df_global = np.diff(full.offsets.astype(np.int64))
    for lo, hi in [(0, 1000), (1000, 1001), (1001, 3000)]:
        keep = (doc >= lo) & (doc < hi)
        part = build_postings(doc[keep] - lo, term[keep], hi - lo, V, doc_len[lo:hi], df_global=df_global, n_docs_global=n,
                              avgdl_global=int(doc_len.sum()) / n)
        ref = full.shard(lo, hi)
        assert np.array_equal(part.offsets, ref.offsets) and np.array_equal(part.doc_ids, ref.doc_ids)
        assert np.array_equal(part.impacts, ref.impacts)
