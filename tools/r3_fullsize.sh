#!/bin/bash
# Round 3: the BASELINE configurations at their size.  tools/r3_fullsize.sh TAG which...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/full_$TAG; mkdir -p $O
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],1), 'solves/s', int(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],2), 'launches', d['roofline']['launches'], 'frac', round(d['roofline']['frac'],3), d['roofline']['bound'], 'iters', d.get('eigenfree_iterations_mean') and round(d['eigenfree_iterations_mean'],1), 'bad', d.get('nonzero_status_points'), 'parity', d.get('parity_sample_max_rel'), d['config']['workload'][:160])"; }
for what in "$@"; do
  case $what in
    mini)   timeout -k 10 300 python bench.py --workload C2-mini --level-slab 4 --steps 2 --warmup 1 --cpu-seconds 3 2>$O/err_mini.log | tee $O/bench_mini_slab4.json | j mini-slab4 || tail -5 $O/err_mini.log
            timeout -k 10 300 python bench.py --workload C2-mini --level-slab 5 --state-slab --steps 2 --warmup 1 --cpu-seconds 3 2>$O/err_mini2.log | tee $O/bench_mini_slab5s.json | j mini-slab5-state || tail -5 $O/err_mini2.log
            timeout -k 10 300 python bench.py --workload C3-mini --level-slab 2 --state-slab --steps 2 --warmup 1 --cpu-seconds 3 2>$O/err_mini3.log | tee $O/bench_c3mini_slab2s.json | j c3mini-slab2-state || tail -5 $O/err_mini3.log ;;
    C3)     timeout -k 10 900 python bench.py --workload C3 --level-slab 20 --steps 2 --warmup 1 --cpu-seconds 8 2>$O/err_C3.log | tee $O/bench_C3.json | j C3 || tail -8 $O/err_C3.log ;;
    C4)     timeout -k 10 900 python bench.py --workload C4-gpu --level-slab 2 --steps 2 --warmup 1 --cpu-seconds 8 2>$O/err_C4.log | tee $O/bench_C4-gpu.json | j C4-gpu || tail -8 $O/err_C4.log ;;
    C4f)    timeout -k 10 900 python bench.py --workload C4-gpu --lists fused --no-torch-lists --steps 2 --warmup 1 2>$O/err_C4f.log | tee $O/bench_C4-gpu_fused.json | j C4-gpu-fused || tail -8 $O/err_C4f.log ;;
    C4p)    timeout -k 10 900 python bench.py --workload C4-gpu --lists pipeline --list-gb 24 --no-torch-lists --steps 2 --warmup 1 2>$O/err_C4p.log | tee $O/bench_C4-gpu_pipeline.json | j C4-gpu-pipeline || tail -8 $O/err_C4p.log ;;
    C4l)    timeout -k 10 900 python bench.py --workload C4-gpu --lists pipeline --list-gb 24 --no-torch-lists --max-nobs 100 --steps 2 --warmup 1 2>$O/err_C4l.log | tee $O/bench_C4-gpu_maxnobs100.json | j C4-gpu-maxnobs100 || tail -8 $O/err_C4l.log ;;
    C4k)    timeout -k 10 900 python bench.py --workload C4h-k100 --lists pipeline --list-gb 24 --no-torch-lists --max-nobs 100 --steps 2 --warmup 1 2>$O/err_C4k.log | tee $O/bench_C4h-k100_maxnobs100.json | j C4h-k100-maxnobs100 || tail -8 $O/err_C4k.log ;;
    C5)     timeout -k 10 900 python bench.py --workload C5-gpu --level-slab 4 --state-slab --steps 2 --warmup 1 --cpu-seconds 8 2>$O/err_C5.log | tee $O/bench_C5-gpu.json | j C5-gpu || tail -8 $O/err_C5.log ;;
  esac
done
