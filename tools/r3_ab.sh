#!/bin/bash
# A/B of a library variant (make VARIANT=name ...) against the production build on the staged workloads, alternating, in one gpurun
# call: tools/r3_ab.sh VARIANT [workloads...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$1; shift
WL=${@:-C3-slab C2-slab-k100 C5-slab}
j() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'solves/s', int(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'bad', d.get('nonzero_status_points'), 'parity', d.get('parity_sample_max_rel'))"; }
for rep in 1 2; do for w in $WL; do
  for lib in base $V; do
    if [ $lib = base ]; then unset LETKF_AMD_LIB; else export LETKF_AMD_LIB=$GRAFT_REPO_ROOT/scale-letkf_amd/lib/libletkf_amd_$V.so; fi
    timeout -k 10 300 python3 bench.py --workload $w --steps 5 --warmup 2 --cpu-seconds 2 2>/dev/null | j "$lib $w"
  done
done; done
