#!/bin/bash
# Kernel statistics of one configs[4] step (run on the GPU box from the repo root):  tools/profile_configE.sh r03
#   rocprofv3 --kernel-trace --stats of bench.py --config E  -> gpurun_out/<tag>_configE_kernel_stats.csv
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r04}
O=gpurun_out
mkdir -p $O
rm -rf $O/${TAG}_configE_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_configE_stats -- python bench.py --config E --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-elbo-check > $O/${TAG}_configE_stats.json 2> $O/${TAG}_configE_stats.err
rc=$?
echo "stats exit $rc"
[ $rc -eq 0 ] || { echo "the profiled run did not exit cleanly"; tail -5 $O/${TAG}_configE_stats.err; exit 1; }
f=$(find $O/${TAG}_configE_stats -name "*kernel_stats.csv" | head -1)
[ -s "$f" ] || { echo "stats failed"; tail -5 $O/${TAG}_configE_stats.err; exit 1; }
cp "$f" $O/${TAG}_configE_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/${TAG}_configE_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step (23 steps): %.1f us" % (tot / 23e3))
for r in rows[:40]:
    print("%-60s calls %6s avg %9.1f us  per step %8.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 23e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
