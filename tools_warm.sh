#!/bin/bash
mkdir -p gpurun_out
for G in 2 4 8 16; do
  echo "== C2 WAVE_GRID=$G"
  LETKF_AMD_WAVE_GRID=$G timeout -k 10 400 python bench.py --workload C2 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['jacobi_sweeps_mean'])"
done
