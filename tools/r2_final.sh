#!/bin/bash
# Round-2 record of the final build: GPU test suite, headline bench with baselines + parity, kernel stats, PMC passes
# (FETCH / WRITE / SQ), bench variants, the staged path (k > 100).  Usage: tools/r2_final.sh TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r02}
O=gpurun_out/final_$TAG
mkdir -p $O
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],2), 'solves/s', int(d['value']), 'solve-only', int(d['solve_only_solves_per_s']), 'kernel_ms', round(d['roofline']['kernel_ms'],2), 'frac', round(d['roofline']['frac'],4), d['roofline']['bound'], 'sweeps', round(d.get('jacobi_sweeps_mean'),2), 'cheb-deg', d.get('chebyshev_degree_mean') and round(d.get('chebyshev_degree_mean'),1), 'bad', d.get('nonzero_status_points'), 'n_mean', d['config']['workload'].split('mean')[1][:7])"; }
echo "== pytest -m gpu"; timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
echo "== tools/profile.sh"; tools/profile.sh $TAG > $O/profile.log 2>&1; tail -12 $O/profile.log | cut -c1-300
echo "== tools/pmc_sq.sh"; tools/pmc_sq.sh $TAG > $O/sq.log 2>&1; tail -12 $O/sq.log
echo "== variants"
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --ensval correlated 2>/dev/null | tee $O/bench_c2_correlated.json | j C2-correlated
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --max-nobs 100 2>/dev/null | tee $O/bench_c2_maxnobs100.json | j C2-maxnobs100
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-search-in-step 2>/dev/null | tee $O/bench_c2_solveonly.json | j C2-solve-only
for w in C2-mini C2-mini-k20 C2-mini-k100 C2-slab-k100 C2-cols-k100 C2-disc C2-mini-disc C2-mini-sparse C4-slab C4-mini C3-mini C3-slab C5-slab; do
  timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds 4 2>/dev/null | tee $O/bench_$w.json | j $w
done
echo "== staged kernels"
tools/r2_prof.sh C3-slab ${TAG}_c3slab | grep "letkf::" | cut -c1-150
tools/r2_prof.sh C5-slab ${TAG}_c5slab | grep "letkf::" | cut -c1-150
tools/r2_prof.sh C4-slab ${TAG}_c4slab | grep "letkf::" | cut -c1-150
tools/r2_prof_args.sh ${TAG}_c2_maxnobs100 --max-nobs 100 | grep "letkf::" | cut -c1-150
echo "== strong scaling, N = 1 (one tile = the whole domain through the subdomain pipeline)"
timeout -k 10 300 python bench.py --scaling strong --steps 3 --warmup 1 2>/dev/null | tee $O/bench_c2_strong_n1.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 strong N=1', 'ms/step', round(d['ms_per_step'],2), 'solves/s', int(d['value']), d['config']['workload'][:120])"
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --state-layout member 2>/dev/null | tee $O/bench_c2_member.json | j C2-member-layout
echo "== bench_largek"; timeout -k 10 500 python bench_largek.py 2>&1 | tail -8
echo "== cycle"; timeout -k 10 300 python bench_cycle.py 2>&1 | tail -3
