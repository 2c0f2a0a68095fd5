#!/bin/bash
# tuning helper: rebuild with different LDS/DPP row splits for the odd Jacobi steps and time C2-mini
mkdir -p gpurun_out
for ks in 10 8 6 4; do
  make -C scale-letkf_amd -B CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -DLETKF_KS_NUM=$ks" > /dev/null 2>&1
  timeout -k 10 200 python bench.py --workload C2-mini --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null > gpurun_out/ks_$ks.json
  python - "$ks" <<'PY'
import sys, json
d = json.load(open(f"gpurun_out/ks_{sys.argv[1]}.json"))
print("KS tenths", sys.argv[1], "ms/step", round(d["ms_per_step"], 3), "solves/s", int(d["value"]), flush=True)
PY
done
