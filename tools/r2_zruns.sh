#!/bin/bash
# A/B of the warm-start run direction / length on C2 (loop body only, production library)
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  dir=${spec%%:*}; R=${spec##*:}
  timeout -k 10 300 python bench.py --workload C2 --steps 3 --warmup 1 --no-cpu-baseline --no-search-in-step --warm-runs $dir --warm-run $R 2> gpurun_out/zr_$dir$R.err > gpurun_out/zr_$dir$R.json || { tail -3 gpurun_out/zr_$dir$R.err; exit 1; }
  python - $dir$R <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/zr_{sys.argv[1]}.json"))
print(sys.argv[1], "ms/step", round(d["ms_per_step"],2), "sweeps", round(d.get("jacobi_sweeps_mean"),3), "parity", d.get("parity_sample_max_rel"), flush=True)
PY
done
