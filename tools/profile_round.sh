#!/bin/bash
# Round profile of the benchmarked command (run on the GPU box from the repo root):  tools/profile_round.sh r03
#   1. bench.py (default flags)                                        -> gpurun_out/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same workload           -> gpurun_out/<tag>_kernel_stats.csv (+ _meta.json)
#   3. rocprofv3 --pmc, one counter set per pass, --kernel-trace only  -> gpurun_out/<tag>_pmc_traffic.json, <tag>_pmc_sq.json
# HBM traffic as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate passes, KB units, FETCH_SIZE x2 on gfx950.
# The program itself follows `--` (no env / bash -c hop under the profiler).
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r04}
K=${2:-200}
W=20
O=gpurun_out
mkdir -p $O
timeout -k 10 400 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { echo "bench failed"; tail -5 $O/${TAG}_bench.err; exit 1; }
echo "bench: $(python -c "import json;d=json.load(open('$O/${TAG}_bench.json'));print(d['value'], d['ms_per_step'], d['ms_per_step_repeats'], d['elbo_check'])")"
rm -rf $O/${TAG}_stats
# (every profiled run must END cleanly: round 2's runs died in the HIP runtime's exit handler -- the cooperative queue's teardown behind
#  the profiler's finalisation, DESIGN.md section 6 -- and this script went on regardless; the launch is a plain one now)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python bench.py --steps $K --warmup $W --repeats 1 --no-cpu-baseline --no-elbo-check --no-call-cost > $O/${TAG}_stats.json 2> $O/${TAG}_stats.err
rc=$?
echo "stats exit $rc"
[ $rc -eq 0 ] || { echo "the profiled run did not exit cleanly"; tail -5 $O/${TAG}_stats.err; exit 1; }
f=$(find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1)
[ -s "$f" ] || { echo "stats failed"; tail -5 $O/${TAG}_stats.err; exit 1; }
cp "$f" $O/${TAG}_kernel_stats.csv
python - <<PY
import csv, glob, json
tr = glob.glob("$O/${TAG}_stats/**/*kernel_trace.csv", recursive=True)
launches = []
if tr:
    for r in csv.DictReader(open(tr[0])):
        if "vjf_mega_kernel" in r["Kernel_Name"]:
            launches.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
json.dump({"mega_launches_us": launches, "command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps $K --warmup $W --repeats 1 --no-cpu-baseline --no-elbo-check --no-call-cost",
           "steps_per_launch": $K, "steps_in_all_launches": $K + $W,
           "note": "vjf_mega_kernel: the warm-up steps ($W, in two launches) and ONE launch of the $K timed steps (= MaxNs); AverageNs is over all three launches, TotalDurationNs / ($K + $W) is the time per step",
           "bench_line": json.load(open("$O/${TAG}_stats.json"))}, open("$O/${TAG}_kernel_stats_meta.json", "w"), indent=1)
PY
for set in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  n=$(echo $set | tr ' ' '_')
  rm -rf $O/${TAG}_pmc_$n
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${TAG}_pmc_$n -- python bench.py --steps $K --warmup $W --repeats 1 --no-cpu-baseline --no-elbo-check --no-call-cost > $O/${TAG}_pmc_$n.json 2> $O/${TAG}_pmc_$n.err
  rc=$?
  echo "pmc $set exit $rc"
  [ $rc -eq 0 ] || { echo "the counter pass '$set' did not exit cleanly"; tail -5 $O/${TAG}_pmc_$n.err; exit 1; }
done
python - <<PY
import csv, glob, json, collections
K, W, O, TAG = $K, $W, "$O", "$TAG"
vals = collections.defaultdict(dict)
for d in glob.glob(f"{O}/{TAG}_pmc_*/"):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "vjf_mega_kernel" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].setdefault("v", []).append(float(r["Counter_Value"]))
out = {}
for k, v in vals.items():
    x = sorted(v["v"])           # two dispatches: W warm-up steps (smaller) and K timed steps (larger)
    out[k] = {"dispatches": len(x), "timed_launch": x[-1], "per_step": x[-1] / K}
print(json.dumps(out, indent=1))
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    fe, wr = out["FETCH_SIZE"]["per_step"] * 2.0 * 1024, out["WRITE_SIZE"]["per_step"] * 1024
    json.dump({"bytes_per_step_corrected": fe + wr, "fetch_bytes_per_step_corrected": fe, "write_bytes_per_step": wr, "steps_per_launch": K,
               "raw": {k: out[k] for k in ("FETCH_SIZE", "WRITE_SIZE")},
               "note": f"rocprofv3 --pmc, separate passes, of the {K}-step vjf_mega_kernel launch of bench.py: FETCH_SIZE x2 (gfx950) + WRITE_SIZE, KB = 1024 B, / {K} steps; "
                       "traffic in the bench line = this x steps_per_launch"}, open(f"{O}/{TAG}_pmc_traffic.json", "w"), indent=1)
json.dump(out, open(f"{O}/{TAG}_pmc_sq.json", "w"), indent=1)
PY
find $O -path "*${TAG}_pmc_*" -name "*.csv" -size +5M -delete
find $O -path "*${TAG}_stats*" -name "*.csv" -size +5M -delete
