# Secondary bench lines of a round with the current build (run on the GPU box from the repo root): driver settings, configs A / C / E,
# the three-stream route, hand-offs without acquires, and the per-kernel profile of a configs[4] step.  Outputs under gpurun_out/.
set -o pipefail
O=gpurun_out
python bench.py --steps 20 --warmup 5 > $O/r02_bench_driver_settings.json 2>$O/e1.err
python bench.py --config A > $O/r02_bench_configA.json 2>$O/e2.err
python bench.py --config C > $O/r02_bench_configC.json 2>$O/e3.err
python bench.py --config E --steps 40 --warmup 5 --repeats 3 > $O/r02_bench_configE.json 2>$O/e4.err
python bench.py --streams-route --no-cpu-baseline > $O/r02_bench_streams_route.json 2>$O/e5.err
VJF_HANDOFF_ACQUIRE=0 python bench.py --no-cpu-baseline > $O/r02_bench_sc1_only.json 2>$O/e6.err
for f in driver_settings configA configC configE streams_route sc1_only; do python -c "
import json; d=json.load(open('$O/r02_bench_$f.json')); print('$f', round(d['value']), [round(x*1e3,2) for x in d['ms_per_step_repeats']], d['elbo_check']['ok'] if d.get('elbo_check') else None, round(d['roofline']['frac'],4))"; done
tools/profile_configE.sh r02 > $O/configE_prof.log 2>&1; head -3 $O/configE_prof.log
