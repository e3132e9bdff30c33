#!/usr/bin/env python3
"""Print the kernel timeline of the last few filter steps from a rocprofv3 --kernel-trace CSV
(start / end / duration in us, queue).  usage: timeline.py <kernel_trace.csv> [n_kernels]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "vjf_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 26
sel = rows[-n:]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:8.1f} {e:8.1f} {e - s:6.1f} q{r['Queue_Id']} grid={r['Grid_Size_X']:>7} {r['Kernel_Name'][:44]}")
