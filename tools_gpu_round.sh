#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, rocprof kernel stats.  Output under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
rm -rf gpurun_out/prof && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --breakdown-steps 0 > gpurun_out/bench_prof.json 2> gpurun_out/prof.err; echo "prof exit $?"
find gpurun_out/prof -name "*kernel_stats*.csv" | head -3
f=$(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -12 "$f"
# keep only the small stats files
find gpurun_out/prof -name "*kernel_trace*.csv" -size +20M -delete
