#!/bin/bash
# one-GPU throughput at the per-rank shard sizes of the 2/4/8-GPU runs (no exchange) and with the exchange path forced on one rank
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], "q/s  step", d["ms_per_step"], "ms  scan", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"], "q/step", d["config"]["queries_per_step"], "fallback", d["fallback_queries"], d["finish_work_rank0"])'
for rows in 500000 250000 125000; do
  python bench.py --rows $rows --legs none --no-cpu-baseline --steps 100 2>/dev/null | python -c "$P" "rows=$rows"
done
python bench.py --rows 125000 --legs none --no-cpu-baseline --steps 100 --force-dist 2>/dev/null | python -c "$P" "rows=125000 one-rank-rccl"
python bench.py --legs none --no-cpu-baseline --steps 100 --force-dist 2>/dev/null | python -c "$P" "rows=1M one-rank-rccl"
