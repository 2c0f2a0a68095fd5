#!/bin/bash
# Round 3: the eigen-free stage under realistic spectra -- obs-space spread sweep.
# Usage: tools/r3_spread.sh TAG [workloads...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r03}; shift
WL=${@:-C3-slab C5-slab C2-slab-k100}
O=gpurun_out/spread_$TAG
mkdir -p $O
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'solves/s', int(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],2), 'frac', round(d['roofline']['frac'],3), 'eigenfree', d.get('eigenfree_points'), 'iters mean/max', d.get('eigenfree_iterations_mean') and round(d['eigenfree_iterations_mean'],1), d.get('eigenfree_iterations_max'), 'fallback', d.get('eigenfree_fallback_points'), 'bad', d.get('nonzero_status_points'), 'parity', d.get('parity_sample_max_rel'))"; }
for w in $WL; do
  timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 4 2>$O/err_$w.log | tee $O/bench_${w}_iid.json | j "$w iid" || tail -5 $O/err_$w.log
  for s in 0.8 1.2 1.6 2.4 4.0; do
    timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 4 --ensval correlated --obs-spread $s 2>$O/err_${w}_$s.log | tee $O/bench_${w}_corr$s.json | j "$w corr $s" || tail -5 $O/err_${w}_$s.log
  done
done
