#!/bin/bash
# round 2: the staged path (k > 100): batched letkf_core with T output, and the das loop on the slab workloads
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],2), 'solves/s', int(d['value']), 'sweeps', d.get('jacobi_sweeps_mean'), 'bad', d.get('nonzero_status_points'), 'kernel_ms', d['roofline']['kernel_ms'])"; }
timeout -k 10 500 python bench_largek.py 2>&1 | tail -8 || exit 1
for w in C3-slab C5-slab C3-mini; do
  timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | j $w || exit 1
done
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c3slab -o c3slab -- python bench.py --workload C3-slab --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_c3slab.log 2>&1
find gpurun_out/prof_c3slab -name "*kernel_stats*" | head -1 | xargs -I{} cp {} gpurun_out/r02_c3slab_kernel_stats.csv
head -12 gpurun_out/r02_c3slab_kernel_stats.csv
