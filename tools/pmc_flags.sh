#!/bin/bash
# Diagnostic: instruction counts of the launches WITHOUT an RLS update (vjf_mega_lite_kernel: trial + SGD [+ moments] roles only), from
# which the trial role's own share can be read (rocprofv3 --pmc, one counter set per pass, --kernel-trace only).
#   tools/pmc_flags.sh   -> gpurun_out/r04_pmc_flags.json
O=gpurun_out; K=200; W=20
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for cfg in "infer 1" "infer 0" "warmup 0"; do
  set -- $cfg; fl=$1; nomom=$2
  for cs in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS"; do
    n=${fl}_${nomom}_$(echo $cs | tr ' ' '_')
    rm -rf $O/pmcf_$n
    if [ $nomom = 1 ]; then export VJF_NO_MOMENTS_ROLE=1; else unset VJF_NO_MOMENTS_ROLE; fi
    timeout -k 10 300 rocprofv3 --pmc $cs --kernel-trace --output-format csv -d $O/pmcf_$n -- python bench.py --flags $fl --steps $K --warmup $W --repeats 1 --no-cpu-baseline --no-elbo-check --no-call-cost > $O/pmcf_$n.json 2> $O/pmcf_$n.err || { echo "pass $n failed"; tail -5 $O/pmcf_$n.err; exit 1; }
  done
done
python - <<PY
import csv, glob, json, collections
out = {}
for d in sorted(glob.glob("$O/pmcf_*/")):
    tag = d.rstrip("/").split("pmcf_")[1]
    fl, nomom = tag.split("_")[:2]
    key = f"{fl}{' (no moments role)' if nomom == '1' else ''}"
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "vjf_mega" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in vals.items():
            out.setdefault(key, {})[k] = max(v) / $K
    try:
        out[key]["us_per_step"] = json.load(open(d.rstrip("/") + ".json"))["ms_per_step"] * 1e3
    except Exception as e:
        pass
json.dump({"per_step": out, "note": "rocprofv3 --pmc of bench.py --flags ... --steps $K: the $K-step launch / $K; B = 4096: 128 trial workgroups x 8 wavefronts"}, open("$O/r04_pmc_flags.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -path "*pmcf_*" -name "*.csv" -size +2M -delete
