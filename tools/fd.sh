#!/bin/bash
# Sharded path on ONE rank (bench.py --force-dist), schedule variants: us/step and status bits.
for v in "$@"; do
  r=$(env $v timeout -k 10 200 python bench.py --no-cpu-baseline --breakdown-steps 0 --force-dist 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), d['status_bits'])")
  echo "[${v:-default}] $r"
done
