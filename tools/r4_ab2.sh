#!/bin/bash
# A/B of SEVERAL library variants against the production build in one gpurun call: tools/r4_ab2.sh "v1 v2" "bench args" workloads...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
VS=$1; shift
ARGS=$1; shift
j() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'solves/s', int(d['value']), 'ms/step', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'bad', d.get('nonzero_status_points'), 'sweeps', d.get('jacobi_sweeps_mean'), 'parity', d.get('parity_sample_max_rel'))"; }
for rep in 1 2; do for w in "$@"; do
  for lib in base $VS; do
    if [ $lib = base ]; then unset LETKF_AMD_LIB; else export LETKF_AMD_LIB=$GRAFT_REPO_ROOT/scale-letkf_amd/lib/libletkf_amd_$lib.so; fi
    timeout -k 10 300 python3 bench.py --workload $w --steps 5 --warmup 2 --cpu-seconds 2 $ARGS 2>/dev/null | j "$lib $w"
  done
done; done
