#!/bin/bash
mkdir -p gpurun_out
for R in 1 16; do
  echo "== C2-mini-k100 RUN_LEN=$R"
  LETKF_AMD_RUN_LEN=$R timeout -k 10 400 python bench.py --workload C2-mini-k100 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['jacobi_sweeps_mean'])"
done
