#!/usr/bin/env python3
"""Print this repository's kernels from a rocprofv3 kernel_stats CSV: calls, average and total time."""
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "vjf_" in r["Name"]]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    print(f'{r["Name"][:52]:52s} calls {int(r["Calls"]):6d}  avg_us {float(r["AverageNs"]) / 1e3:9.1f}  total_ms {float(r["TotalDurationNs"]) / 1e6:8.2f}')
