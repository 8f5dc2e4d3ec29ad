#!/bin/bash
# Encoder profiles of a round (run through gpurun): kernel-trace stats and the MFMA-utilisation counters of
# tools/bench_encoder.py (256 x 512 tokens, 24 layers), plus tools/bench_gemm.py's per-layer GEMM table.
# Everything lands in gpurun_out/<tag>_*; copy what is to be judged into profiles/.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_enc -- python3 $R/tools/bench_encoder.py --iters 2 > $O/${TAG}_prof_enc.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${TAG}_pmc_enc -- python3 $R/tools/bench_encoder.py --iters 1 > $O/${TAG}_pmc_enc.log 2>&1 || exit 1
cd $R
cp $(ls $O/${TAG}_prof_enc/*/*_kernel_stats.csv | head -1) $O/${TAG}_encoder_bs256x512_kernel_stats.csv
python3 tools/pmc_mfma_summary.py $(ls $O/${TAG}_pmc_enc/*/*_counter_collection.csv | head -1) $O/${TAG}_pmc_encoder_mfma.json > /dev/null
python3 tools/bench_encoder.py > $O/${TAG}_encoder_bs256x512.json 2>/dev/null
python3 tools/bench_gemm.py --yardstick > $O/${TAG}_gemm_layers.jsonl 2>/dev/null
python3 - <<PY
import json
d = json.load(open("$O/${TAG}_pmc_encoder_mfma.json"))
print("whole forward MFMA util", d["whole_forward_mfma_util"])
for k, v in d["kernels"].items():
    print(k.ljust(40), v["mfma_util"])
print(open("$O/${TAG}_encoder_bs256x512.json").read())
PY
