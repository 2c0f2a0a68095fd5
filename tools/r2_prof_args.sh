#!/bin/bash
# kernel-level breakdown of a bench.py run with arbitrary arguments: tools/r2_prof_args.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_${TAG}_kernel_stats.csv \;
tail -1 $OUT/trace.log | cut -c1-600
head -8 gpurun_out/r02_${TAG}_kernel_stats.csv | cut -c1-200
rm -rf $OUT/trace
