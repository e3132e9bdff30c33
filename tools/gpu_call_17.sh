#!/bin/bash
# Round 3, GPU call 17: where configs C and E stand; config C role timeline
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c17
mkdir -p $O
for c in C E A; do
timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c exit $?"
python -c "
import json; d=json.load(open('$O/bench_$c.json')); print('$c', '%.2f M' % (d['value']/1e6), ['%.2f' % (x*1e3) for x in d['ms_per_step_repeats']], d['roofline']['frac'], d['config'].get('route'))"
done
CFG=C timeout -k 10 200 python tools/mega_stamps.py > $O/stamps_C.txt 2>&1; echo "stamps exit $?"
grep -v amdgpu.ids $O/stamps_C.txt | grep -E "\[3[78]\]" | head -120
