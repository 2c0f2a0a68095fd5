#!/bin/bash
# three points per wave (letkf_trio.hip) against the one-point register kernel, same library: LETKF_OPT_SMALL_K_TRIO = 6 off / on
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
j() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'solves/s', int(d['value']), 'ms/step', round(d['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4), 'bad', d.get('nonzero_status_points'), 'sweeps', d.get('jacobi_sweeps_mean'), 'parity', d.get('parity_sample_max_rel'), d['roofline']['kernel'][:40])"; }
for rep in 1 2; do for w in "$@"; do
  for o in 0 1; do
    timeout -k 10 300 python3 bench.py --workload $w --steps 5 --warmup 2 --cpu-seconds 2 --ctx-option 6=$o 2>/dev/null | j "trio=$o $w"
  done
done; done
