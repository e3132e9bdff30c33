#!/bin/bash
# Round 3, GPU call 12: statistics Gram of the wide route as a split-K GEMM
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c12
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "config_E or few_trials or state_noise_when or split_entry or nonfinite_component_is_dropped or random_configurations" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -6 $O/pytest.log
timeout -k 10 300 python bench.py --config E --no-cpu-baseline > $O/bench_E.json 2> $O/bench_E.err; echo "bench E exit $?"
python -c "
import json; d=json.load(open('$O/bench_E.json')); print('E', '%.3f M' % (d['value']/1e6), ['%.1f' % (x*1e3) for x in d['ms_per_step_repeats']], d['roofline']['frac'], d['elbo_check'])"
rm -rf $O/E_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/E_stats -- python bench.py --config E --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-elbo-check > $O/E_stats.json 2> $O/E_stats.err; echo "rocprof exit $?"
f=$(find $O/E_stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/configE_kernel_stats.csv
find $O -name "*.csv" -size +3M -delete
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/configE_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step (23 steps): %.1f us" % (tot / 23e3))
for r in rows[:16]:
    print("%-64s calls %5s avg %8.1f us  per step %7.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 23e3))
PY
