#!/bin/bash
# Round 3, GPU call 29: kernel trace of the three-stream route (one rank) at config B
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c29
mkdir -p $O
rm -rf $O/trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python bench.py --streams-route --steps 12 --warmup 4 --repeats 1 --no-cpu-baseline --no-elbo-check > $O/trace.out 2> $O/trace.err; echo "trace exit $?"
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_streams.py $F vjf_prepg_kernel > $O/timeline_streams.txt 2>&1; echo "timeline exit $?"
find $O -name "*.csv" -size +3M -delete
cat $O/timeline_streams.txt
