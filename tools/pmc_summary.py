"""Summarise the two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE; separate passes) of tools/seq_search.py into the JSON
bench.py reads for roofline.traffic.  usage: pmc_summary.py <fetch_dir> <write_dir> <out.json> <algorithmic_bytes_per_launch>"""
import csv, glob, json, sys


def collect(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter and ("scan_bf16_kernel" in r["Kernel_Name"] or "scan_split_kernel" in r["Kernel_Name"]) and ", true" in r["Kernel_Name"]]
    # ", true" = the multi-pass form of the template (single-query callers get the ", false" form)
    vals = vals[3:]          # drop the first launches
    return {"launches": len(vals), "mean_kb": sum(vals) / len(vals), "min_kb": min(vals), "max_kb": max(vals)}


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
alg = int(sys.argv[4])
fb, wb = fetch["mean_kb"] * 1024 * 2, write["mean_kb"] * 1024
out = {"kernel": "scan_bf16_kernel<IP, 8 waves, ring 16, multi-pass>, 8 passes per launch (round 3: candidate lists kept by the scan itself)",
       "workload": "1M x 1024 rows (bf16 filter copy, 2.048 GB), 512 queries per launch (tools/seq_search.py: sequential hipidx_search_dev)",
       "commands": ["rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 tools/seq_search.py",
                    "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python3 tools/seq_search.py"],
       "counters": {"FETCH_SIZE": fetch, "WRITE_SIZE": write},
       "corrections": "FETCH_SIZE is in KiB and on gfx950 reports exactly 1/2 of the bytes of a wide coalesced streaming "
                      "read (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE (KiB) is exact for 16-byte streaming stores",
       "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "traffic_bytes_per_launch": fb + wb,
       "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (fb + wb) / alg}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("traffic_bytes_per_launch", "traffic_over_algorithmic")}))
