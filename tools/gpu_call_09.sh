#!/bin/bash
# Round 3, GPU call 9: one collective per chain, exact non-finite replay on the sharded route (opt-in) -- suite + sharded-route benches
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c09
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -8 $O/pytest.log
for v in "--streams-route" "--force-dist"; do
  n=$(echo $v | tr -d '-')
  timeout -k 10 300 python bench.py $v --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err; echo "bench $v exit $?"
  python -c "
import json; d=json.load(open('$O/bench_$n.json')); print('$v', '%.2f M' % (d['value']/1e6), ['%.2f' % (x*1e3) for x in d['ms_per_step_repeats']], d['elbo_check'] and d['elbo_check']['ok'])"
done
