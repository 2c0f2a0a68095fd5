#!/bin/bash
mkdir -p gpurun_out
for DBG in 3 1; do
  echo "== C2 MAX_SWEEP=60 RUN_LEN=16 DBG=$DBG"
  LETKF_AMD_WARM_DBG=$DBG LETKF_AMD_RUN_LEN=16 timeout -k 10 400 python bench.py --workload C2 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['jacobi_sweeps_mean'])"
done
