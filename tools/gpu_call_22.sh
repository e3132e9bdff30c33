#!/bin/bash
# Round 3, GPU call 22: config E, internal streams at the highest priority; streams route of config B too (it uses the same streams)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c22
mkdir -p $O
timeout -k 10 300 python bench.py --config E --no-cpu-baseline > $O/bench_E.json 2> $O/bench_E.err; echo "bench E exit $?"
timeout -k 10 300 python bench.py --streams-route --no-cpu-baseline > $O/bench_streams.json 2> $O/bench_streams.err; echo "bench streams exit $?"
python - <<PY
import json
for f in ("bench_E", "bench_streams"):
    d = json.load(open("$O/%s.json" % f))
    print(f, "%.2f M" % (d["value"] / 1e6), ["%.1f" % (x * 1e3) for x in d["ms_per_step_repeats"]], d["roofline"]["frac"], "enq", d["roofline"].get("host_enqueue_us_per_step"))
PY
