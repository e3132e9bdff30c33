#!/bin/bash
# Round 3, GPU call 23: config E, timing events around the phases of the two-stream route (no profiler)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c23
mkdir -p $O
VJF_DEBUG_TWO_TIMELINE=1 timeout -k 10 300 python bench.py --config E --steps 8 --warmup 4 --repeats 1 --no-cpu-baseline --no-elbo-check > $O/bench_E.json 2> $O/bench_E.err; echo "bench E exit $?"
grep two-timeline $O/bench_E.err | sort -k2 -n | tail -60
