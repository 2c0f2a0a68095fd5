#!/bin/bash
# k = 100 two-wave kernel: phase shares (PROF twin) and the run direction A/B
cd $GRAFT_REPO_ROOT
export LETKF_AMD_LIB=$GRAFT_REPO_ROOT/scale-letkf_amd/lib/libletkf_amd_prof.so
for dir in x z; do
  timeout -k 10 300 python bench.py --workload ${1:-C2-slab-k100} --steps 2 --warmup 1 --no-cpu-baseline --no-search-in-step --warm-runs $dir 2> gpurun_out/k100_$dir.err > gpurun_out/k100_$dir.json || { tail -3 gpurun_out/k100_$dir.err; exit 1; }
  python - $dir <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/k100_{sys.argv[1]}.json"))
print(sys.argv[1], d["config"]["workload"][:60], "ms/step", round(d["ms_per_step"],2), "solves/s", int(d["value"]), "sweeps", round(d.get("jacobi_sweeps_mean"),3), flush=True)
PY
  grep "letkf prof" gpurun_out/k100_$dir.err | tail -1 | cut -c1-220
done
