#!/bin/bash
# Round 3, GPU call 19: kernel trace of config E on the two-stream route
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c19
mkdir -p $O
rm -rf $O/trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python bench.py --config E --steps 8 --warmup 4 --repeats 1 --no-cpu-baseline --no-elbo-check > $O/trace.out 2> $O/trace.err; echo "trace exit $?"
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_streams.py $F > $O/timeline_E.txt 2>&1; echo "timeline exit $?"
find $O -name "*.csv" -size +3M -delete
cat $O/timeline_E.txt
