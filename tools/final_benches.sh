#!/bin/bash
# Secondary bench lines of a round with the current build (run on the GPU box from the repo root):  tools/final_benches.sh r03
# driver settings, configs A / C / E (E also in the one-stream order), the three-stream route with and without RCCL on one rank,
# hand-offs with acquires, the fixed cost of a call, the role timeline, the per-kernel profile of a configs[4] step.
# Outputs under gpurun_out/.  Any step that does not exit cleanly ends the script.
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r04}
O=gpurun_out
mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/${TAG}_bench_$name.json 2> $O/${TAG}_bench_$name.err || { echo "bench $name failed"; tail -5 $O/${TAG}_bench_$name.err; exit 1; }; }
run driver_settings --steps 20 --warmup 5
run configA --config A
run configC --config C
run configE --config E --steps 40 --warmup 5 --repeats 3
run configE_one_stream --config E --steps 40 --warmup 5 --repeats 3 --no-overlap --no-cpu-baseline
run streams_route --streams-route --no-cpu-baseline
run force_dist_one_rank --force-dist --no-cpu-baseline
run force_dist_one_rank_one_collective --force-dist --collectives 1 --no-cpu-baseline
VJF_HANDOFF_ACQUIRE=1 run with_acquires --no-cpu-baseline
run D1 --config D1
run flags_warmup --flags warmup --no-cpu-baseline
run flags_infer --flags infer --no-cpu-baseline
run flags_sgd-only --flags sgd-only --no-cpu-baseline
VJF_NO_MOMENTS_ROLE=1 run flags_warmup_no_moments_role --flags warmup --no-cpu-baseline
VJF_NO_MOMENTS_ROLE=1 run flags_infer_no_moments_role --flags infer --no-cpu-baseline
for f in driver_settings configA configC configE configE_one_stream streams_route force_dist_one_rank force_dist_one_rank_one_collective with_acquires D1 flags_warmup flags_infer flags_sgd-only flags_warmup_no_moments_role flags_infer_no_moments_role; do python -c "
import json; d=json.load(open('$O/${TAG}_bench_$f.json')); print('$f', round(d['value']), [round(x*1e3,2) for x in d['ms_per_step_repeats']], d['elbo_check']['ok'] if d.get('elbo_check') else None, round(d['roofline']['frac'],4))"; done
timeout -k 10 200 python tools/call_cost.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_call_cost.txt || { echo "call_cost failed"; exit 1; }
cat $O/${TAG}_call_cost.txt
timeout -k 10 200 python tools/mega_stamps.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_roles_timeline.txt || { echo "mega_stamps failed"; exit 1; }
FLAGS=warmup timeout -k 10 200 python tools/mega_stamps.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_roles_timeline_warmup.txt || { echo "mega_stamps (warm-up) failed"; exit 1; }
FLAGS=infer timeout -k 10 200 python tools/mega_stamps.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_roles_timeline_infer.txt || { echo "mega_stamps (infer) failed"; exit 1; }
VJF_DEBUG_TWO_TIMELINE=1 timeout -k 10 300 python bench.py --config E --steps 8 --warmup 4 --repeats 1 --no-cpu-baseline --no-elbo-check > /dev/null 2> $O/${TAG}_two_tl.err || { echo "config E timeline failed"; exit 1; }
grep two-timeline $O/${TAG}_two_tl.err | sort -k2 -n | tail -36 > $O/${TAG}_configE_phase_timeline.txt
tools/profile_configE.sh $TAG > $O/${TAG}_configE_prof.log 2>&1 || { echo "config E profile failed"; tail -5 $O/${TAG}_configE_prof.log; exit 1; }
head -12 $O/${TAG}_configE_prof.log
