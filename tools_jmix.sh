#!/bin/bash
mkdir -p gpurun_out
for jm in 1 3 5; do
  LETKF_AMD_JMIX=$jm timeout -k 10 200 python bench.py --workload ${1:-C2-mini} --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null > gpurun_out/jmix_$jm.json
  python - "$jm" <<'PY'
import sys, json
d = json.load(open(f"gpurun_out/jmix_{sys.argv[1]}.json"))
print("jmix", sys.argv[1], "ms/step", round(d["ms_per_step"], 3), "solves/s", int(d["value"]), flush=True)
PY
done
