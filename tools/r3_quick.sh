#!/bin/bash
# quick look at the staged workloads: tools/r3_quick.sh TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-q}; O=gpurun_out/quick_$TAG; mkdir -p $O
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'solves/s', int(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],2), 'frac', round(d['roofline']['frac'],3), 'iters', d.get('eigenfree_iterations_mean') and round(d['eigenfree_iterations_mean'],1), 'fallback', d.get('eigenfree_fallback_points'), 'bad', d.get('nonzero_status_points'), 'parity', d.get('parity_sample_max_rel'))"; }
for w in C3-slab C2-slab-k100 C5-slab C3-mini; do
  timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 3 2>$O/err_$w.log | tee $O/bench_${w}_iid.json | j "$w iid" || tail -5 $O/err_$w.log
  timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 3 --ensval correlated --obs-spread 2.4 2>$O/err_${w}_24.log | tee $O/bench_${w}_corr2.4.json | j "$w corr 2.4" || tail -5 $O/err_${w}_24.log
done
