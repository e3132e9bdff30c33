#!/bin/bash
# Round 3, GPU call 31: three-stream route, in-kernel wait for the statistics
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c31
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "sharded or three_streams or held_at_their or workspace_contents or fake_world" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline $2 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 exit $?"
  python - <<PY
import json
d = json.load(open("$O/bench_$1.json"))
print("$1", "%.2f M" % (d["value"] / 1e6), ["%.1f" % (x * 1e3) for x in d["ms_per_step_repeats"]], d["roofline"]["frac"], "enq", d["roofline"].get("host_enqueue_us_per_step"))
PY
}
run streams --streams-route
run forcedist --force-dist
rm -rf $O/trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python bench.py --streams-route --steps 12 --warmup 4 --repeats 1 --no-cpu-baseline --no-elbo-check > $O/trace.out 2> $O/trace.err; echo "trace exit $?"
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_streams.py $F vjf_prepg_kernel > $O/timeline_streams.txt 2>&1; echo "timeline exit $?"
find $O -name "*.csv" -size +3M -delete
cat $O/timeline_streams.txt
