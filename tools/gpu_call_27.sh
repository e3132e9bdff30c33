#!/bin/bash
# Round 3, GPU call 27: resident column loop with write-through stores and sc1 loads
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c27
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "own_stream or nonfinite_component_is_dropped or config_E or multi_launch or weights_nearly or few_trials or rls_failure or random_configurations" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -6 $O/pytest.log
run() {
  timeout -k 10 300 python bench.py --config E --no-cpu-baseline $2 > $O/bench_$1.json 2> $O/bench_$1.err; echo "bench $1 exit $?"
  python - <<PY
import json
d = json.load(open("$O/bench_$1.json"))
print("$1", "%.2f M" % (d["value"] / 1e6), ["%.1f" % (x * 1e3) for x in d["ms_per_step_repeats"]], d["roofline"]["frac"], "enq", d["roofline"].get("host_enqueue_us_per_step"))
PY
}
run loop
run loop_onestream --no-overlap
VJF_RLS_COLUMN_LAUNCHES=1 run percol
VJF_DEBUG_TWO_TIMELINE=1 timeout -k 10 300 python bench.py --config E --steps 8 --warmup 4 --repeats 1 --no-cpu-baseline --no-elbo-check > $O/tl.json 2> $O/tl.err; echo "timeline exit $?"
grep two-timeline $O/tl.err | sort -k2 -n | tail -20
