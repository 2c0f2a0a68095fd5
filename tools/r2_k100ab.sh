#!/bin/bash
# k = 100 / 80 / 72 on the slab grids: two-wave kernel (loop body only)
cd $GRAFT_REPO_ROOT
for w in "$@"; do
  timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-search-in-step 2> gpurun_out/k100ab.err > gpurun_out/k100ab.json || { tail -3 gpurun_out/k100ab.err; exit 1; }
  python - <<'PY'
import json
d=json.load(open("gpurun_out/k100ab.json"))
print(d["config"]["workload"][:44], "ms/step", round(d["ms_per_step"],2), "solves/s", int(d["value"]), "sweeps", round(d.get("jacobi_sweeps_mean"),3), "bad", d.get("nonzero_status_points"), d["roofline"]["kernel"][:50], flush=True)
PY
done
