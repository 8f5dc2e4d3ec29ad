"""Summarise a rocprofv3 --pmc FETCH_SIZE --kernel-trace run of tools/bench_hybrid.py: HBM fetch (KiB, as reported) and mean
duration per kernel.  usage: pmc_hybrid_summary.py <pmc_dir> <out.json>"""
import csv, glob, json, sys
from collections import defaultdict

TARGETS = ("scan_bf16_kernel", "scan_split_kernel", "fin_kernel", "taat_tile_kernel", "merge_packed_loop_kernel", "merge_packed_kernel",
           "rrf_kernel", "exhaustive_kernel", "merge_wave_kernel")


def short(name):
    for t in TARGETS:
        if t in name:
            return t
    return None


d = sys.argv[1]
f = glob.glob(d + "/*/*counter_collection.csv")[0]
fetch = defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        fetch[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
dur = defaultdict(list)
kt = glob.glob(d + "/*/*kernel_trace.csv")
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[short(r["Kernel_Name"])].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
out = {"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 tools/bench_hybrid.py",
       "workload": "configs[2]: 1M chunks, 148 M postings, 256 queries per launch, 6 terms per query (449,804 postings = 3.6 MB requested per query)",
       "note": "FETCH_SIZE in KiB per launch as reported; on gfx950 it reads half the bytes of wide (dwordx4) streaming loads "
               "(MI355X_MICROARCH.md), so the dense scan row is x2 in bytes; the BM25 stream uses 4-byte loads, for which the factor is not "
               "established: its HBM traffic lies between the raw figure and twice that",
       "kernels": {}}
for k, v in fetch.items():
    if k is None:
        continue
    v2 = v[2:] if len(v) > 4 else v
    e = {"launches": len(v2), "fetch_kib_per_launch": round(sum(v2) / len(v2), 1)}
    if dur.get(k):
        dd = dur[k][2:] if len(dur[k]) > 4 else dur[k]
        e["avg_duration_us"] = round(sum(dd) / len(dd), 1)
        e["raw_TBps"] = round(e["fetch_kib_per_launch"] * 1024 / (e["avg_duration_us"] * 1e-6) / 1e12, 3)
    if "taat_tile" in k:
        e["posting_bytes_requested_per_launch"] = 449804 * 8 * 256
        e["requested_over_fetched_raw"] = round(e["posting_bytes_requested_per_launch"] / (e["fetch_kib_per_launch"] * 1024), 2)
        k = "taat_tile_kernel"
    out["kernels"][k] = e
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["kernels"].get("taat_tile_kernel")))
