#!/bin/bash
# kernel-level breakdown of a bench.py workload: tools/r2_prof.sh <workload> <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-C3-slab}; TAG=${2:-x}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_${TAG}_kernel_stats.csv \;
tail -2 $OUT/trace.log
head -8 gpurun_out/r02_${TAG}_kernel_stats.csv | cut -c1-220
rm -rf $OUT/trace
