"""Diagnostic: per-kernel timeline of one steady-state step from a rocprofv3 --kernel-trace csv (start / end per kernel and queue):
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --config E --steps 8 --warmup 4 --repeats 1 ...
    python tools/trace_streams.py DIR/.../*_kernel_trace.csv [first_kernel_substring]
prints the kernels between the second-to-last and the last launch of `vjf_wide_in_kernel` (or the given name), times relative to it."""
import csv, sys, glob
f = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "vjf_wide_in_kernel"
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
marks = [i for i, r in enumerate(rows) if key in r[2]]
a, b = marks[-3], marks[-2]
t0 = rows[a][0]
qs = sorted({r[3] for r in rows[a:b]})
print(f"step of {(rows[b][0] - t0) / 1e3:.1f} us; queues {qs}")
for s, e, n, q, st in rows[a:b + 1]:
    col = qs.index(q) if q in qs else 0
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {'    ' * col}q{q} {n}")
