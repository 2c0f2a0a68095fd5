#!/bin/bash
mkdir -p gpurun_out
for L in torch fused; do
for MS in 60 0; do
  echo "== C2 lists=$L MAX_SWEEP=$MS"
  LETKF_AMD_MAX_SWEEP=$MS timeout -k 10 400 python bench.py --workload C2 --steps 2 --warmup 1 --no-cpu-baseline --lists $L 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['jacobi_sweeps_mean'])"
done
done
