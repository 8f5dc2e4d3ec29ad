#!/bin/bash
# Collects the round's judged artefacts on the GPU box (run through gpurun): bench line, rocprofv3 kernel stats of the same
# command, the sequential per-kernel profile and the two PMC passes of the scan.  Everything lands in gpurun_out/<tag>_*;
# copy what is to be judged into profiles/.
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R && python3 bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err || exit 1
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_bench -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --legs none > $O/${TAG}_prof_bench.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_seq -- python3 $R/tools/seq_search.py > $O/${TAG}_prof_seq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_fetch -- python3 $R/tools/seq_search.py > $O/${TAG}_pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_write -- python3 $R/tools/seq_search.py > $O/${TAG}_pmc_write.log 2>&1 || exit 1
# the SURVEY 8(d)-priced scan (fp32 rows streamed): kernel stats of the same bench command in q64 mode
export HIPRAG_SCAN_MODE=q64
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_q64 -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --legs none > $O/${TAG}_prof_q64.log 2>&1 || exit 1
unset HIPRAG_SCAN_MODE
cd $R
cp $(ls $O/${TAG}_prof_q64/*/*_kernel_stats.csv | head -1) $O/${TAG}_bench_q64_kernel_stats.csv
python3 tools/pmc_summary.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_scan.json 16384000000
# hybrid legs (configs[2]): kernel stats + HBM fetch of the current BM25 / dense top-50 kernels
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_hybrid -- python3 $R/tools/bench_hybrid.py > $O/${TAG}_prof_hybrid.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_hybrid -- python3 $R/tools/bench_hybrid.py > $O/${TAG}_pmc_hybrid.log 2>&1 || exit 1
cd $R
cp $(ls $O/${TAG}_prof_hybrid/*/*_kernel_stats.csv | head -1) $O/${TAG}_hybrid_config2_kernel_stats.csv
python3 tools/pmc_hybrid_summary.py $O/${TAG}_pmc_hybrid $O/${TAG}_pmc_hybrid_fetch.json
for t in prof_bench prof_seq; do
  cp $(ls $O/${TAG}_$t/*/*_kernel_stats.csv | head -1) $O/${TAG}_${t}_kernel_stats.csv
done
python3 - <<PY
import csv
for t in ("prof_bench", "prof_seq"):
    print("==", t)
    for r in list(csv.DictReader(open("$O/${TAG}_%s_kernel_stats.csv" % t)))[:10]:
        print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), "us", r["Percentage"])
PY
cat $O/${TAG}_bench_n1.json
