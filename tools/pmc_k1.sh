#!/bin/bash
# SQ counters of the trial kernel (serial schedule: --pmc serialises kernels).  One pass per counter group.
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rm -rf gpurun_out/pmck_$i
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmck_$i -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --breakdown-steps 0 --serial-schedule > gpurun_out/pmck_$i.json 2> gpurun_out/pmck_$i.err
  echo "group $i exit $?"
done
python - <<'PY'
import csv, glob, collections
for i in range(1, 5):
    f = glob.glob(f"gpurun_out/pmck_{i}/**/*counter_collection.csv", recursive=True)
    if not f: print("no file", i); continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "vjf_trial_mfma" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        print(f"{k:32s} n={len(v):3d} avg={sum(v)/len(v):14.1f}  min={min(v):14.1f} max={max(v):14.1f}")
PY
