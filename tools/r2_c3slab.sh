#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --workload ${1:-C3-slab} --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/c3s.err > gpurun_out/c3s.json || { tail -3 gpurun_out/c3s.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/c3s.json"))
print(d["config"]["workload"][:50], "ms/step", round(d["ms_per_step"],2), "solves/s", int(d["value"]), "sweeps", round(d.get("jacobi_sweeps_mean"),3), "bad", d.get("nonzero_status_points"), "parity", d.get("parity_sample_max_rel"), flush=True)
PY
