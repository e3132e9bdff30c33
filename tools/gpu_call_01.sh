#!/bin/bash
# Round 3, GPU call 1: exit-fault probes under rocprofv3, baseline bench lines, the new parity tests, the pinned hard cases.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c01
mkdir -p $O
echo "== exit probes"
for mode in torch load step coop close; do
  rm -rf $O/probe_$mode
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/probe_$mode -- python tools/exit_probe.py $mode $O/maps_$mode.txt > $O/probe_$mode.out 2> $O/probe_$mode.err
  echo "probe $mode exit $?"
done
echo "== bench under rocprofv3 (as tools/profile_round.sh runs it), maps dumped by BENCH_MAPS"
rm -rf $O/probe_bench
VJF_BENCH_MAPS=$O/maps_bench.txt timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/probe_bench -- python bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-elbo-check > $O/probe_bench.out 2> $O/probe_bench.err
echo "probe bench exit $?"
find $O -name "*.csv" -size +2M -delete
echo "== plain bench"
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default exit $?"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench driver exit $?"
python - <<PY
import json
for f in ("bench_default", "bench_driver"):
    try:
        d = json.load(open("$O/%s.json" % f))
        print(f, d["value"], d["ms_per_step"], d["ms_per_step_repeats"], d["roofline"]["frac"], d["elbo_check"])
    except Exception as e:
        print(f, "unreadable", e)
PY
echo "== new tests"
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "large_k_major or gemm_dispatcher or last_step_without or config_E_at_bench_size or bench_line_contract" > $O/pytest_new.log 2>&1; echo "pytest new exit $?"; tail -5 $O/pytest_new.log
echo "== pinned hard cases"
timeout -k 10 600 python tools/fuzz_parity.py 0 -1 > $O/hard.log 2>&1; echo "hard exit $?"; grep -v amdgpu.ids $O/hard.log | cut -c1-260
for c in 39 185; do timeout -k 10 200 python tools/fuzz_parity.py 0 -1 $c > $O/hard_$c.log 2>&1; done
grep -h "w_chol\|w_pchol\|w_mean\|w_prec" $O/hard_39.log $O/hard_185.log | tail -40
