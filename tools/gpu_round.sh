#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, rocprof kernel stats.  Output under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | cut -c1-200 | head -20
if [ "$1" != "quick" ]; then
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log | cut -c1-200
fi
timeout -k 10 600 python bench.py ${BENCH_ARGS} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"; cut -c1-1800 gpurun_out/bench.json; tail -3 gpurun_out/bench.err | cut -c1-300
rm -rf gpurun_out/prof && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --breakdown-steps 0 > gpurun_out/bench_prof.json 2> gpurun_out/prof.err; echo "prof exit $?"
f=$(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'vjf_' in r['Name']]
w = csv.writer(open('gpurun_out/kernel_stats_vjf.csv', 'w'))
w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
for r in rows:
    w.writerow([r['Name'], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
    print(r['Name'][:60], r['Calls'], 'avg_us=%.1f' % (float(r['AverageNs']) / 1e3), r['Percentage'] + '%')
PY
find gpurun_out/prof -name "*kernel_trace*.csv" -delete
find gpurun_out/prof -name "*kernel_stats*.csv" -delete
