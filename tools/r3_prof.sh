#!/bin/bash
# kernel breakdown of one bench workload: tools/r3_prof.sh TAG <bench.py args...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/bench.json 2> $O/err.log
f=$(find $O -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/kernel_stats.csv && head -12 $f | cut -c1-220
find $O -type f ! -name "kernel_stats.csv" ! -name "bench.json" ! -name "err.log" -delete
