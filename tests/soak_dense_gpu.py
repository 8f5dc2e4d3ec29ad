"""Long randomised parity soak of the dense index against the fp64 oracle (not collected by pytest; run on a GPU box:
`python tests/soak_dense_gpu.py <seed> <cases>`).  Shapes, k, batch sizes, metric, scan mode, un-normalised rows, duplicates,
zero rows / queries and tight clusters are drawn at random; every case is checked id for id like tests/test_dense_gpu.py."""
import os, sys, numpy as np
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'intool-rag_amd')); sys.path.insert(0,os.path.join(R,'tests'))
from oracle import hybrid_oracle as ho
import test_dense_gpu as T
from hiprag import HipFlatIndex
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 777)
bad=0
for case in range(int(sys.argv[2]) if len(sys.argv)>2 else 150):
    n = int(rng.choice([1, 2, 31, 32, 33, 64, 500, 4097, 9000, 30011, 65537]))
    d = int(rng.choice([4, 8, 100, 128, 384, 1000, 1024]))
    k = int(rng.choice([1, 5, 10, 31, 50, 57, 64, 128, 129]))
    nq = int(rng.choice([1, 2, 63, 64, 65, 200, 300, 513]))
    metric = [ho.METRIC_IP, ho.METRIC_L2][case % 2]
    mode = ["bf16", "bf16", "bf16", "q64"][int(rng.integers(4))]
    os.environ["HIPRAG_SCAN_MODE"]=mode
    # few scan workgroups -> many blocks per wave at these small sizes: several publish chunks per wave and pass, short last
    # chunks of every length, values tested many blocks after their pass ended (what a 1M-row index does with the whole grid)
    spare = int(rng.choice([0, 0, 192, 240, 252, 254]))
    x = ho.synthetic_vectors(n, d, seed=5000 + case); q = ho.synthetic_queries(nq, d, seed=6000 + case)
    if case % 3 == 0: x *= rng.uniform(0.1, 30.0, size=(n, 1)).astype(np.float32)
    if case % 4 == 1 and n > 40: x[rng.integers(0, n, size=20)] = x[3]
    if case % 5 == 2: x[n // 2] = 0; q[0] = 0
    if case % 7 == 3 and n > 100:   # a tight cluster: many near-ties
        c = x[5].copy(); idx = rng.integers(0, n, size=min(n, 300)); x[idx] = c + 1e-4 * rng.standard_normal((len(idx), d)).astype(np.float32)
    ix = HipFlatIndex(d, metric); ix.add(x[: n // 2]); ix.add(x[n // 2:]); ix.set_spare_cus(spare)
    try:
        T._check(ix, x, q, k, metric)
    except AssertionError as e:
        bad+=1; print(f"FAIL case {case}: n={n} d={d} k={k} nq={nq} metric={metric} mode={mode}: {str(e)[:200]}", flush=True)
    if case % 25 == 0: print("case", case, "ok so far, bad =", bad, flush=True)
print("done, failures:", bad)
