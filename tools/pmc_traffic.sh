#!/bin/bash
# HBM traffic per filter step from the L2 fabric counters (MI355X_MICROARCH.md §HBM):
# two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only; FETCH_SIZE is doubled
# (gfx950 reports half the bytes of wide coalesced reads), WRITE_SIZE is taken as is.  Units: KB.
# --pmc serialises kernels: the run uses --serial-schedule (the pipelined schedule's kernels, one stream).
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --breakdown-steps 0 --serial-schedule > gpurun_out/pmc_$c.json 2> gpurun_out/pmc_$c.err
  echo "pmc $c exit $?"
done
python - <<'PY'
import csv, glob, json, collections
NSTEP = 35                               # --steps 30 --warmup 5: every vjf_ kernel of the run belongs to one of them
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counter file for", c); continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") == c and "vjf_" in r["Kernel_Name"]:
            per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    out[c] = {k: {"dispatches": len(v), "avg_KB": sum(v) / len(v), "KB_per_step": sum(v) / NSTEP} for k, v in per.items()}
names = sorted(set(out.get("FETCH_SIZE", {})) | set(out.get("WRITE_SIZE", {})))
tot = 0.0
for k in names:
    if "vjf_aux_kernel" in k or "triclean" in k:
        continue                      # run once per API call, not per step
    fe = out.get("FETCH_SIZE", {}).get(k, {}).get("KB_per_step", 0.0) * 2.0     # gfx950 correction
    wr = out.get("WRITE_SIZE", {}).get(k, {}).get("KB_per_step", 0.0)
    nd = out.get("FETCH_SIZE", {}).get(k, {}).get("dispatches", 0)
    print(f"{k:36s} {nd / NSTEP:4.1f} launches/step  fetch(corrected) {fe:10.1f} KB/step  write {wr:10.1f} KB/step")
    tot += fe + wr
out["bytes_per_step_corrected"] = tot * 1024
out["note"] = "sum over all dispatches of the run / 35 steps; FETCH_SIZE x2 (gfx950), WRITE_SIZE as is; KB = 1024 B"
print("HBM bytes per step (corrected):", tot * 1024)
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
PY
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +5M -delete
