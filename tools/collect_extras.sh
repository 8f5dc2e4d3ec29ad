#!/bin/bash
# The round's secondary artefacts (run through gpurun after collect_profiles.sh): per-rank shard sizes on one GPU, the one-query
# path, the config-5-shaped end-to-end run, and the A/B probes behind DESIGN 3.3 / 5 (scan beside its tails; hybrid legs on
# disjoint CUs).  Everything lands in gpurun_out/<tag>_*.
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
bash tools/shard_sizes.sh > $O/${TAG}_shard_sizes.txt 2>&1
python3 tools/bench_query_path.py > $O/${TAG}_query_path.json 2> /dev/null
python3 tools/bench_e2e.py > $O/${TAG}_e2e.json 2> /dev/null
for rows in 125000 1000000; do
  ROWS=$rows SPARES=0,48 STEPS=60 timeout -k 10 300 python3 tools/scan_split_probe.py 2> /dev/null | grep rows >> $O/${TAG}_scan_split.jsonl
done
# the same arrangement after the process has used k other streams first (hardware-queue assignment): the start gate at work
for k in 2 3 5; do
  ROWS=125000 SPARES=48 STEPS=60 ASIDE_ONLY=1 USE_SKIPPED=1 SKIP_STREAMS=$k timeout -k 10 300 python3 tools/scan_split_probe.py 2> /dev/null | grep rows | sed "s/^{/{\"streams_used_first\": $k, /" >> $O/${TAG}_scan_split.jsonl
done
SPARES=0,64,96 timeout -k 10 400 python3 tools/hybrid_split_probe.py 2> /dev/null | grep "{" > $O/${TAG}_hybrid_split.jsonl
DOCS=200000 SPARES=0,96 timeout -k 10 400 python3 tools/hybrid_split_probe.py 2> /dev/null | grep "{" | sed 's/^{/{"docs": 200000, /' >> $O/${TAG}_hybrid_split.jsonl
cat $O/${TAG}_shard_sizes.txt | cut -c1-140
tail -c 600 $O/${TAG}_query_path.json; tail -c 400 $O/${TAG}_e2e.json
cat $O/${TAG}_scan_split.jsonl | cut -c1-330
cat $O/${TAG}_hybrid_split.jsonl | cut -c1-420
