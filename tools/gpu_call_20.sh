#!/bin/bash
# Round 3, GPU call 20: config E, forward half of t+1 enqueued before the update's launches
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c20
mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "own_stream or nonfinite_component_is_dropped or config_E" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest.log
timeout -k 10 300 python bench.py --config E --no-cpu-baseline > $O/bench_E.json 2> $O/bench_E.err; echo "bench E exit $?"
python - <<PY
import json
for f in ("bench_E",):
    d = json.load(open("$O/%s.json" % f))
    print(f, "%.2f M" % (d["value"] / 1e6), ["%.1f" % (x * 1e3) for x in d["ms_per_step_repeats"]], d["roofline"]["frac"], "enq", d["roofline"].get("host_enqueue_us_per_step"))
PY
rm -rf $O/trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python bench.py --config E --steps 8 --warmup 4 --repeats 1 --no-cpu-baseline --no-elbo-check > $O/trace.out 2> $O/trace.err; echo "trace exit $?"
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_streams.py $F > $O/timeline_E.txt 2>&1; echo "timeline exit $?"
find $O -name "*.csv" -size +3M -delete
grep -v rlsc_col $O/timeline_E.txt
