export TMPDIR=/tmp
O=gpurun_out
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES" "SQ_IFETCH SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $set | tr ' ' '_')
  rm -rf $O/r04i_pmc_$n
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/r04i_pmc_$n -- python bench.py --steps 200 --warmup 20 --repeats 1 --no-cpu-baseline --no-elbo-check --no-call-cost > $O/r04i_pmc_$n.json 2> $O/r04i_pmc_$n.err
  echo "pmc $set exit $?"
done
python - <<PY
import csv, glob, collections
vals = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r04i_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "vjf_mega_kernel" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(vals.items()):
    print(k, max(v), "per step", max(v) / 200)
PY
find gpurun_out -path "*r04i_pmc_*" -name "*.csv" -size +2M -delete
