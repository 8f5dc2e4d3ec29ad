#!/usr/bin/env python3
"""Print the tail of a rocprofv3 kernel-trace CSV as a timeline: start / end (us, relative) and name of the last N kernels."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"]
    for cut in ("hiprag::(anonymous namespace)::", "void ", "hiprag::"):
        name = name.replace(cut, "")
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:10.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:10.1f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f}  q{r.get("Queue_Id", "?")}  grid {r.get("Grid_Size", "?"):>9}  {name[:70]}')
