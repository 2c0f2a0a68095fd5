#!/bin/bash
# production library against its PROF twin on the workloads with empty points (C2-disc, C2-mini-noobs), twice each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],2), 'kernel_ms', round(d['roofline']['kernel_ms'],2), 'solves/s', int(d['value']))"; }
P=$PWD/scale-letkf_amd/lib/libletkf_amd_prof.so
for w in ${WL:-C2-disc C2-mini-noobs}; do
 for rep in 1 2; do
  timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | j $w-prod-$rep
  LETKF_AMD_LIB=$P timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | j $w-prof-$rep
 done
done
