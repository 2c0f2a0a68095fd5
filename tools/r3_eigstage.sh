#!/bin/bash
# the fallback measured on its own: every staged point through the eigen stage.  tools/r3_eigstage.sh TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-e}; O=gpurun_out/eig_$TAG; mkdir -p $O
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'solves/s', int(d['value']), 'ms/step', round(d['ms_per_step'],1), 'bad', d.get('nonzero_status_points'), 'parity', d.get('parity_sample_max_rel'))"; }
for w in C5-slab C3-slab C3-mini; do
  timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 --cpu-seconds 3 --eigen-stage-only 2>$O/err_$w.log | tee $O/bench_${w}_iid.json | j "$w iid eig-only" || tail -5 $O/err_$w.log
  timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 --cpu-seconds 3 --eigen-stage-only --ensval correlated --obs-spread 2.4 2>$O/err_${w}_24.log | tee $O/bench_${w}_corr2.4.json | j "$w corr 2.4 eig-only" || tail -5 $O/err_${w}_24.log
done
