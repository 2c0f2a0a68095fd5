#!/bin/bash
# profiling helper: time the das kernel with the Jacobi sweep cap lowered (LETKF_AMD_MAX_SWEEP knob)
mkdir -p gpurun_out
for ms in 60 0 1 2; do
  LETKF_AMD_MAX_SWEEP=$ms timeout -k 10 200 python bench.py --workload ${1:-C2-mini} --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null > gpurun_out/phase_$ms.json
  python - "$ms" <<'PY'
import sys, json
ms = sys.argv[1]
d = json.load(open(f"gpurun_out/phase_{ms}.json"))
print("maxsweep", ms, "ms/step", round(d["ms_per_step"], 3), "solves/s", int(d["value"]), "sweeps", d["jacobi_sweeps_mean"], flush=True)
PY
done
