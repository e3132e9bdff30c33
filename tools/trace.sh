#!/bin/bash
# Kernel-trace timeline of a short bench run (gpurun_out/timeline.txt).  Extra args go to bench.py.
export TMPDIR=/tmp
rm -rf gpurun_out/trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --breakdown-steps 0 "$@" > gpurun_out/trace.json 2> gpurun_out/trace.err || exit 1
f=$(find gpurun_out/trace -name "*kernel_trace*.csv" | head -1)
python tools/timeline.py "$f" ${TIMELINE_N:-26} | tee gpurun_out/timeline.txt
rm -rf gpurun_out/trace
