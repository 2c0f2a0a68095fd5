#!/bin/bash
# A/B of a library variant against the production build on the headline workload (and its sweeps), alternating, in one gpurun
# call: tools/r3_ab_c2.sh VARIANT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$1
j() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],2), 'kernel_ms', round(d['roofline']['kernel_ms'],2), 'sweeps', d.get('jacobi_sweeps_mean'), 'bad', d.get('nonzero_status_points'), 'parity', d.get('parity_sample_max_rel'))"; }
for rep in 1 2; do
  for lib in base $V; do
    if [ $lib = base ]; then unset LETKF_AMD_LIB; else export LETKF_AMD_LIB=$GRAFT_REPO_ROOT/scale-letkf_amd/lib/libletkf_amd_$V.so; fi
    timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --cpu-seconds 2 2>/dev/null | j "$lib C2"
  done
done
for lib in base $V; do
  if [ $lib = base ]; then unset LETKF_AMD_LIB; else export LETKF_AMD_LIB=$GRAFT_REPO_ROOT/scale-letkf_amd/lib/libletkf_amd_$V.so; fi
  timeout -k 10 300 python3 bench.py --workload C2-mini-sparse --steps 6 --warmup 2 --cpu-seconds 2 2>/dev/null | j "$lib C2-mini-sparse"
  timeout -k 10 300 python3 bench.py --workload C2-slab-k100 --eigen-stage-only --steps 4 --warmup 1 --cpu-seconds 2 2>/dev/null | j "$lib k100-eig-only"
done
