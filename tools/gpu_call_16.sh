#!/bin/bash
# Round 3, GPU call 16: Gram sum stored in 16-byte pieces both ways; sc1 loads of G in the Cholesky loop
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c16
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -8 $O/pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench default exit $?"
python - <<PY
import json
for f in ("bench_default",):
    try:
        d = json.load(open("$O/%s.json" % f))
        print(f, "%.2f M" % (d["value"] / 1e6), ["%.2f" % (x * 1e3) for x in d["ms_per_step_repeats"]], d["roofline"]["frac"], d["elbo_check"])
    except Exception as e:
        print(f, "unreadable", e)
PY
timeout -k 10 200 python tools/mega_stamps.py > $O/stamps.txt 2>&1; echo "stamps exit $?"
grep -v amdgpu.ids $O/stamps.txt | grep -E "\[3[78]\]" | head -120
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench driver exit $?"
python -c "
import json; d=json.load(open('$O/bench_driver.json')); print('driver', '%.2f M' % (d['value']/1e6), ['%.2f' % (x*1e3) for x in d['ms_per_step_repeats']], 'enq us/step', d['roofline']['host_enqueue_us_per_step'])"
timeout -k 10 200 python tools/call_cost.py 2>&1 | grep -v amdgpu.ids
