#!/bin/bash
# records of the k <= 20 workloads on letkf_trio.hip: bench lines (CPU baseline, parity) + rocprofv3 kernel stats of C2-k20
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_trio_rec; mkdir -p $O
timeout -k 10 300 python3 bench.py --workload C1 --steps 10 --warmup 2 --cpu-seconds 8 > $O/bench_C1.json 2> $O/err_C1.log || tail -5 $O/err_C1.log
timeout -k 10 300 python3 bench.py --workload C2-k20 --steps 5 --warmup 2 --cpu-seconds 8 > $O/bench_C2-k20.json 2> $O/err_C2k20.log || tail -5 $O/err_C2k20.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 bench.py --workload C2-k20 --steps 3 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/c2k20_kernel_stats.csv \;
head -6 $O/c2k20_kernel_stats.csv
for f in C1 C2-k20; do python3 -c "
import json; d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', int(d['value']), 'solves/s', round(d['ms_per_step'],3), 'ms kernel', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],4), d['roofline']['bound'], 'parity', d['parity_sample_max_rel'], 'cpu', d['cpu_baseline'] and int(d['cpu_baseline']['value']), d['roofline']['kernel'][:30])"; done
