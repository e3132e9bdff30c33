#!/bin/bash
# Round 3, GPU call 2: is the exit fault the cooperative queue under rocprofv3 (no torch, no libvjf)?  plain launch A/B; hard cases vs the fp32-LAPACK oracle
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03_c02
mkdir -p $O
for mode in plain coop; do
  rm -rf $O/repro_$mode
  timeout -k 10 60 rocprofv3 --kernel-trace --output-format csv -d $O/repro_$mode -- tools/coop_exit_repro $mode > $O/repro_$mode.out 2> $O/repro_$mode.err
  echo "standalone $mode under rocprofv3: exit $?"; cat $O/repro_$mode.out
done
tools/coop_exit_repro coop; echo "standalone coop, no profiler: exit $?"
rm -rf $O/probe_plainlaunch
VJF_DEBUG_PLAIN_LAUNCH=1 timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/probe_plainlaunch -- python tools/exit_probe.py coop $O/maps_plainlaunch.txt > $O/probe_plainlaunch.out 2> $O/probe_plainlaunch.err
echo "probe coop with a plain launch under rocprofv3: exit $?"; tail -2 $O/probe_plainlaunch.out
find $O -name "*.csv" -size +2M -delete
echo "== driver-settings bench, cooperative vs plain launch"
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-elbo-check > $O/bench_coop_$i.json 2> $O/bench_coop_$i.err; echo "coop exit $?"
VJF_DEBUG_PLAIN_LAUNCH=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-elbo-check > $O/bench_plain_$i.json 2> $O/bench_plain_$i.err; echo "plain exit $?"
done
python - <<PY
import json
for f in ("bench_coop_1", "bench_plain_1", "bench_coop_2", "bench_plain_2"):
    try:
        d = json.load(open("$O/%s.json" % f))
        print(f, "%.2f M" % (d["value"] / 1e6), ["%.2f" % (x * 1e3) for x in d["ms_per_step_repeats"]], "enqueue us/step %.2f" % d["roofline"]["host_enqueue_us_per_step"])
    except Exception as e:
        print(f, "unreadable", e)
PY
echo "== pinned hard cases against the oracle with an fp32 LAPACK factorisation"
timeout -k 10 600 python tools/fuzz_parity.py 0 -1 > $O/hard.log 2>&1; echo "hard exit $?"; grep -v amdgpu.ids $O/hard.log | cut -c1-300
echo "== role timeline"
timeout -k 10 200 python tools/mega_stamps.py > $O/stamps.txt 2>&1; echo "stamps exit $?"; grep -v amdgpu.ids $O/stamps.txt | head -60
