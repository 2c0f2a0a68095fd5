#!/bin/bash
# Round-3 record of one build: tools/r3_final.sh TAG part   (part A: tests, headline + counters, variants; part B: spread
# sweeps, staged counters, the BASELINE configurations at their size, cycle).  Everything under gpurun_out/final_TAG*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r03}; PART=${2:-A}
O=gpurun_out/final_$TAG
mkdir -p $O
j() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],2), 'solves/s', int(d['value']), 'solve-only', int(d['solve_only_solves_per_s']), 'kernel_ms', round(d['roofline']['kernel_ms'],2), 'frac', round(d['roofline']['frac'],4), d['roofline']['bound'], 'sweeps', round(d.get('jacobi_sweeps_mean'),2), 'iters', d.get('eigenfree_iterations_mean') and round(d.get('eigenfree_iterations_mean'),1), 'fallback', d.get('eigenfree_fallback_points'), 'bad', d.get('nonzero_status_points'), 'parity', d.get('parity_sample_max_rel'), 'n_mean', d['config']['workload'].split('mean')[1][:7])"; }
if [ $PART = A ]; then
  echo "== pytest -m gpu"; timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
  echo "== tools/profile.sh"; tools/profile.sh $TAG > $O/profile.log 2>&1; tail -12 $O/profile.log | cut -c1-300
  echo "== tools/pmc_sq.sh"; tools/pmc_sq.sh $TAG > $O/sq.log 2>&1; tail -12 $O/sq.log
  echo "== variants"
  timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --ensval correlated 2>/dev/null | tee $O/bench_c2_correlated.json | j C2-correlated
  timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --ensval correlated --obs-spread 2.4 2>/dev/null | tee $O/bench_c2_correlated_spread2.4.json | j C2-correlated-2.4
  timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --max-nobs 100 2>/dev/null | tee $O/bench_c2_maxnobs100.json | j C2-maxnobs100
  timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-search-in-step 2>/dev/null | tee $O/bench_c2_solveonly.json | j C2-solve-only
  timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --state-layout member 2>/dev/null | tee $O/bench_c2_member.json | j C2-member-layout
  for w in C2-mini C2-mini-k20 C2-k20 C1 C2-mini-k100 C2-slab-k100 C2-cols-k100 C2-disc C2-mini-disc C2-mini-sparse C4-slab C4-mini C3-mini C3-slab C5-slab; do
    timeout -k 10 300 python3 bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 4 2>/dev/null | tee $O/bench_$w.json | j $w
  done
  echo "== strong scaling, N = 1"; timeout -k 10 300 python3 bench.py --scaling strong --steps 3 --warmup 1 2>/dev/null | tee $O/bench_c2_strong_n1.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 strong N=1', 'ms/step', round(d['ms_per_step'],2), 'solves/s', int(d['value']))"
  echo "== cycle"; timeout -k 10 300 python3 bench_cycle.py 2>&1 | tail -3
else
  echo "== spread sweeps"; tools/r3_spread.sh $TAG C3-slab C5-slab C2-slab-k100
  echo "== staged counters"
  tools/r3_pmc.sh ${TAG}_c3slab --workload C3-slab > $O/pmc_c3slab.out 2>&1; tail -2 $O/pmc_c3slab.out | head -1
  tools/r3_pmc.sh ${TAG}_c3slab24 --workload C3-slab --ensval correlated --obs-spread 2.4 > $O/pmc_c3slab24.out 2>&1
  tools/r3_pmc.sh ${TAG}_k100 --workload C2-slab-k100 > $O/pmc_k100.out 2>&1
  tools/r3_pmc.sh ${TAG}_c5slab --workload C5-slab > $O/pmc_c5slab.out 2>&1
  tools/r3_pmc.sh ${TAG}_c2k20 --workload C2-k20 > $O/pmc_c2k20.out 2>&1
  for t in c3slab c3slab24 k100 c5slab c2k20; do python3 -c "
import json
d=json.load(open('gpurun_out/pmc_${TAG}_$t/summary.json'))
print('$t', round(d['bench']['value']), d['bench']['roofline']['frac'])
for k,e in d['kernels'].items():
    if e.get('pct',0)>5: print('   ', k, 'ms', round(e['avg_ms'],3), 'traffic GB', round(e.get('traffic_bytes_per_launch',0)/1e9,2), 'TB/s', round(e.get('traffic_TB_per_s',0),2), 'wait', round(e.get('wait_inst_any_frac_of_wave_cycles',0),2), 'valu', round(e.get('valu_busy_frac',0),2))
"; done
  echo "== full size"; tools/r3_fullsize.sh $TAG C3 C4p C4l C5
fi
