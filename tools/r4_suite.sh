#!/bin/bash
# round 4: the whole GPU suite (runtime messages on the log: pytest.ini --capture=sys) + the contract bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_suite_${1:-a}; mkdir -p $O
echo "== pytest -m gpu"; timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
echo "== bench"; timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 > $O/bench_c2.json 2> $O/bench_c2.err; python3 -c "
import json; d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]); print('C2', round(d['ms_per_step'],1), 'ms', int(d['value']), 'solves/s kernel_ms', round(d['roofline']['kernel_ms'],1), 'frac', round(d['roofline']['frac'],4), 'parity', d['parity_sample_max_rel'], d['parity_lists'], 'cpu', d['cpu_baseline'] and int(d['cpu_baseline']['value']))"
