#!/bin/bash
# Round 3, GPU call 24: config E, the RLS streams on compute units of their own (CU masks): 0 (none) / 32 / 48 / 64
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c24
mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "own_stream or nonfinite_component_is_dropped or config_E" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest.log
for r in 0 32 48 64; do
VJF_TWO_CUS=$r timeout -k 10 300 python bench.py --config E --no-cpu-baseline > $O/bench_E_$r.json 2> $O/bench_E_$r.err; echo "bench E cus=$r exit $?"
python - <<PY
import json
d = json.load(open("$O/bench_E_$r.json"))
print("cus=$r", "%.2f M" % (d["value"] / 1e6), ["%.1f" % (x * 1e3) for x in d["ms_per_step_repeats"]], d["roofline"]["frac"], "enq", d["roofline"].get("host_enqueue_us_per_step"))
PY
done
VJF_DEBUG_TWO_TIMELINE=1 timeout -k 10 300 python bench.py --config E --steps 8 --warmup 4 --repeats 1 --no-cpu-baseline --no-elbo-check > $O/tl.json 2> $O/tl.err; echo "timeline exit $?"
grep two-timeline $O/tl.err | sort -k2 -n | tail -26
