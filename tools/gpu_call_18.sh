#!/bin/bash
# Round 3, GPU call 18: multi-launch RLS update on its own stream (config E): tests, bench with / without
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c18
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "own_stream or nonfinite_component_is_dropped or config_E or multi_launch or weights_nearly or few_trials" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -8 $O/pytest.log
timeout -k 10 300 python bench.py --config E --no-cpu-baseline > $O/bench_E.json 2> $O/bench_E.err; echo "bench E exit $?"
timeout -k 10 300 python bench.py --config E --no-cpu-baseline --no-overlap > $O/bench_E_one.json 2> $O/bench_E_one.err; echo "bench E one-stream exit $?"
python - <<PY
import json
for f in ("bench_E", "bench_E_one"):
    try:
        d = json.load(open("$O/%s.json" % f))
        print(f, "%.2f M" % (d["value"] / 1e6), ["%.1f" % (x * 1e3) for x in d["ms_per_step_repeats"]], d["roofline"]["frac"], d.get("elbo_check"))
    except Exception as e:
        print(f, "unreadable", e)
PY
