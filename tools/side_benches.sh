#!/bin/bash
# Side measurements quoted in DESIGN.md / README.md (GPU box): analysis cycle by stage, search-inclusive analysis,
# other ensemble sizes, large-k paths.  Usage: ./tools/side_benches.sh > gpurun_out/side.log
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],2), 'solves/s', int(d['value']), 'sweeps', d.get('jacobi_sweeps_mean'), 'bad', d.get('nonzero_status_points'))"; }
timeout -k 10 300 python bench_cycle.py 2>&1 | tail -4 || exit 1
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --lists columns --search-in-step 2>/dev/null | j C2-search-in-step || exit 1
for w in C2-mini C2-mini-k20 C2-mini-k100 C2-mini-noobs; do
  timeout -k 10 200 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | j $w || exit 1
done
timeout -k 10 400 python bench_largek.py 2>&1 | tail -8
