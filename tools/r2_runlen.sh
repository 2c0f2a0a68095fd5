#!/bin/bash
# A/B of the warm-start run length on C2 (PROF twin's LETKF_AMD_RUN_LEN knob; loop body only)
cd $GRAFT_REPO_ROOT
for R in ${@:-16 32 64}; do
  LETKF_AMD_LIB=$GRAFT_REPO_ROOT/scale-letkf_amd/lib/libletkf_amd_prof.so LETKF_AMD_RUN_LEN=$R timeout -k 10 300 python bench.py --workload C2 --steps 3 --warmup 1 --no-cpu-baseline --no-search-in-step 2> gpurun_out/runlen_$R.err > gpurun_out/runlen_$R.json || { tail -3 gpurun_out/runlen_$R.err; exit 1; }
  python - $R <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/runlen_{sys.argv[1]}.json"))
print("run", sys.argv[1], "ms/step", round(d["ms_per_step"],2), "sweeps", d.get("jacobi_sweeps_mean"), flush=True)
PY
  grep "letkf prof" gpurun_out/runlen_$R.err | tail -1 | cut -c1-200
done
