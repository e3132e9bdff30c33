#!/bin/bash
# Round 3, GPU call 25: config E two-stream route: stream priority x short-workgroup GEMM for Z
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c25
mkdir -p $O
run() {
  timeout -k 10 300 python bench.py --config E --no-cpu-baseline $2 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 exit $?"
  python - <<PY
import json
d = json.load(open("$O/bench_$1.json"))
print("$1", "%.2f M" % (d["value"] / 1e6), ["%.1f" % (x * 1e3) for x in d["ms_per_step_repeats"]], d["roofline"]["frac"])
PY
}
run base
VJF_DEBUG_STREAM_PRIO=1 run prio
VJF_DEBUG_NO_GEMM128=1 run no128
VJF_DEBUG_NO_GEMM128=1 VJF_DEBUG_STREAM_PRIO=1 run no128_prio
VJF_DEBUG_NO_GEMM128=1 run no128_onestream --no-overlap
