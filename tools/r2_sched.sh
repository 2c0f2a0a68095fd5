#!/bin/bash
# A/B of the wave kernel's run scheduling: C2 and C2-disc with the production library (dynamic) and, in the PROF twin, dynamic
# against the static dealing (LETKF_AMD_STATIC_SCHED).  Usage: tools/r2_sched.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/sched; mkdir -p $O
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],2), 'kernel_ms', round(d['roofline']['kernel_ms'],2), 'solves/s', int(d['value']), 'sweeps', round(d.get('jacobi_sweeps_mean'),3), 'bad', d.get('nonzero_status_points'), 'parity', d.get('parity_sample_max_rel'))"; }
for w in C2 C2-disc; do
  timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 4 2>$O/$w.err | tee $O/$w.json | j $w-dynamic
  LETKF_AMD_LIB=$PWD/scale-letkf_amd/lib/libletkf_amd_prof.so timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>$O/$w.prof_dyn.err | tee $O/$w.prof_dyn.json | j $w-prof-dynamic
  LETKF_AMD_STATIC_SCHED=1 LETKF_AMD_LIB=$PWD/scale-letkf_amd/lib/libletkf_amd_prof.so timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>$O/$w.prof_static.err | tee $O/$w.prof_static.json | j $w-prof-static
done
