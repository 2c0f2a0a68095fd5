#!/bin/bash
# A/B of library variants (make -C scale-letkf_amd VARIANT=name EXTRA=...) inside ONE gpurun call, alternating runs.
# Usage: tools/ab_variants.sh "<bench args>" variant1 variant2 ...   ("-" = the production library)
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then unset LETKF_AMD_LIB; else export LETKF_AMD_LIB=$GRAFT_REPO_ROOT/scale-letkf_amd/lib/libletkf_amd_$v.so; fi
  timeout -k 10 300 python bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/ab.err > gpurun_out/ab.json || { tail -3 gpurun_out/ab.err; exit 1; }
  python - "$v" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab.json"))
print(sys.argv[1].ljust(10), d["config"]["workload"][:40], "ms/step", round(d["ms_per_step"],3), "kernel_ms", round(d["roofline"]["kernel_ms"],3), "solves/s", int(d["value"]), "sweeps", round(d.get("jacobi_sweeps_mean") or 0,3), "bad", d.get("nonzero_status_points"), flush=True)
PY
done
done
