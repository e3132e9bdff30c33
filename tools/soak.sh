#!/bin/bash
# Soak: the bench workload several times (status bits must stay 0, the ELBO of equal runs equal), then the GPU tests twice.
for k in 500 500 3000 1500 500; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --breakdown-steps 0 --steps $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps', d['steps'], 'us/step', round(d['ms_per_step']*1e3,2), 'status', d['status_bits'], 'elbo', d['elbo'])"
done
for i in 1 2; do timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1; done
