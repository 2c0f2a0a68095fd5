#!/bin/bash
# Round 4: the full-size records of the routes whose round-3 numbers carried no parity: the checker's lists now
# come from the oracle's obs_local (bench.py cpu_baseline), so --no-torch-lists / --max-nobs runs are checked like every other.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/full_$TAG; mkdir -p $O
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step', round(d['ms_per_step'],1), 'solves/s', int(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],2), 'frac', round(d['roofline']['frac'],3), d['roofline']['bound'], 'bad', d.get('nonzero_status_points'), 'parity', d.get('parity_sample_max_rel'), d.get('parity_lists'), 'cpu', d['cpu_baseline'] and int(d['cpu_baseline']['value']), d['config']['workload'][:120])"; }
for what in "$@"; do
  case $what in
    C4p)    timeout -k 10 1100 python bench.py --workload C4-gpu --lists pipeline --list-gb 24 --no-torch-lists --steps 2 --warmup 1 --cpu-seconds 8 2>$O/err_C4p.log | tee $O/bench_C4-gpu_pipeline.json | j C4-gpu-pipeline || tail -8 $O/err_C4p.log ;;
    C4l)    timeout -k 10 1100 python bench.py --workload C4-gpu --lists pipeline --list-gb 24 --no-torch-lists --max-nobs 100 --steps 2 --warmup 1 --cpu-seconds 8 2>$O/err_C4l.log | tee $O/bench_C4-gpu_maxnobs100.json | j C4-gpu-maxnobs100 || tail -8 $O/err_C4l.log ;;
    C4k)    timeout -k 10 1100 python bench.py --workload C4h-k100 --lists pipeline --list-gb 24 --no-torch-lists --max-nobs 100 --steps 2 --warmup 1 --cpu-seconds 8 2>$O/err_C4k.log | tee $O/bench_C4h-k100_maxnobs100.json | j C4h-k100-maxnobs100 || tail -8 $O/err_C4k.log ;;
    C4ku)   timeout -k 10 1100 python bench.py --workload C4h-k100 --lists pipeline --list-gb 24 --no-torch-lists --steps 1 --warmup 1 --cpu-seconds 8 2>$O/err_C4ku.log | tee $O/bench_C4h-k100_unlimited.json | j C4h-k100-unlimited || tail -8 $O/err_C4ku.log ;;
    C2l)    timeout -k 10 600 python bench.py --max-nobs 100 --steps 3 --warmup 1 --cpu-seconds 8 2>$O/err_C2l.log | tee $O/bench_c2_maxnobs100.json | j C2-maxnobs100 || tail -8 $O/err_C2l.log ;;
    C2lp)   timeout -k 10 600 python bench.py --max-nobs 100 --lists pipeline --steps 3 --warmup 1 --cpu-seconds 8 2>$O/err_C2lp.log | tee $O/bench_c2_maxnobs100_pipeline.json | j C2-maxnobs100-pipeline || tail -8 $O/err_C2lp.log ;;
    C1)     timeout -k 10 300 python bench.py --workload C1 --steps 10 --warmup 2 --cpu-seconds 8 2>$O/err_C1.log | tee $O/bench_C1.json | j C1 || tail -8 $O/err_C1.log ;;
  esac
done
