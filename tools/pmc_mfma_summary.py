#!/usr/bin/env python3
"""MFMA utilisation per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace
--output-format csv` run (one row per dispatch and counter).

    util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)

SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs and counts 16 cycles per v_mfma_f32_16x16x32_bf16 (32 per 32x32x16:
MI355X_MICROARCH.md, per-instruction constants) -- check: the F = 4096 GEMM of 131072 rows issues 2^26 MFMAs and reads
2^30.  GRBM_GUI_ACTIVE is summed over the 8 XCDs.  At 2.4 GHz, 100 % = 1024 SIMDs x 16384 flop / 16 cycles = 2.52 PFLOP/s,
the bf16 dense peak the roofline fractions are quoted against.
usage: pmc_mfma_summary.py <counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- "
                      "python3 tools/bench_encoder.py --iters 1",
           "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)", "kernels": {}}
    tot_busy = tot_cyc = 0.0
    for name, c in agg.items():
        if "hiprag" not in name or "GRBM_GUI_ACTIVE" not in c:
            continue
        n = len(c["GRBM_GUI_ACTIVE"])
        busy = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0]))
        cyc = sum(c["GRBM_GUI_ACTIVE"]) / 8.0
        short = name.replace("hiprag::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        out["kernels"][short] = {"dispatches": n, "mfma_busy_simd_cycles_per_dispatch": busy / n,
                                 "active_cycles_per_dispatch": cyc / n, "mfma_util": round(busy / (cyc * 1024.0), 4)}
        tot_busy += busy
        tot_cyc += cyc
    out["whole_forward_mfma_util"] = round(tot_busy / (tot_cyc * 1024.0), 4)
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
