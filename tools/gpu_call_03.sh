#!/bin/bash
# Round 3, GPU call 3: the whole GPU suite on the plain-launch build + rocprofv3 exit codes + the hard cases
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c03
mkdir -p $O
rm -rf $O/probe_bench
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/probe_bench -- python bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-elbo-check > $O/probe_bench.out 2> $O/probe_bench.err
echo "bench under rocprofv3 exit $?"
find $O -name "*.csv" -size +2M -delete
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -8 $O/pytest.log
