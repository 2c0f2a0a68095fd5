#!/bin/bash
# k = 100 (and 64 / 80): two-wave register kernel against the staged path (PROF twin knob LETKF_AMD_STAGED_MIN_K)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],2), 'solves/s', int(d['value']), 'solve-only', int(d['solve_only_solves_per_s']), 'sweeps', round(d.get('jacobi_sweeps_mean'),2), 'bad', d.get('nonzero_status_points'), d['roofline']['kernel'][:60])"; }
export LETKF_AMD_LIB=$GRAFT_REPO_ROOT/scale-letkf_amd/lib/libletkf_amd_prof.so
for w in C2-slab-k100; do
  timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | j "$w wave" || exit 1
  LETKF_AMD_STAGED_MIN_K=63 timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | j "$w staged" || exit 1
done
