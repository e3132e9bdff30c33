#!/bin/bash
# A/B of schedule variants on ONE box (boxes differ by several %): tools/ab.sh "ENV1=1" "ENV2=1 ENV3=1" ...  ("" = default)
for rep in 1 2 3; do
  for v in "$@"; do
    r=$(env $v timeout -k 10 200 python bench.py --no-cpu-baseline --breakdown-steps 0 --steps 1000 --warmup 50 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step']*1e3,2))")
    echo "rep $rep [${v:-default}] $r us/step"
  done
done
