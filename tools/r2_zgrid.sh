#!/bin/bash
cd $GRAFT_REPO_ROOT
for G in "$@"; do
  LETKF_AMD_LIB=$GRAFT_REPO_ROOT/scale-letkf_amd/lib/libletkf_amd_prof.so LETKF_AMD_WAVE_GRID=$G timeout -k 10 300 python bench.py --workload C2 --steps 3 --warmup 1 --no-cpu-baseline --no-search-in-step 2> gpurun_out/zg_$G.err > gpurun_out/zg_$G.json || { tail -3 gpurun_out/zg_$G.err; exit 1; }
  python - $G <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/zg_{sys.argv[1]}.json"))
print("grid x", sys.argv[1], "ms/step", round(d["ms_per_step"],2), "sweeps", round(d.get("jacobi_sweeps_mean"),3), flush=True)
PY
done
timeout -k 10 400 python bench.py > gpurun_out/zfull.json 2> gpurun_out/zfull.err; cat gpurun_out/zfull.json | cut -c1-1500
