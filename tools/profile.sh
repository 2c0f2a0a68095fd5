#!/bin/bash
# Round profile of the headline bench (needs a GPU): bench line with CPU baseline, rocprofv3 kernel stats, and the two
# PMC passes (FETCH_SIZE, WRITE_SIZE) in their own runs.  Usage: ./tools/profile.sh v6   -> gpurun_out/prof_v6/
set -o pipefail
TAG=${1:-vX}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 500 python bench.py --steps 3 --warmup 1 > $OUT/bench_c2.json 2> $OUT/bench.err || exit 1
tail -c 2500 $OUT/bench_c2.json; echo
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1 || exit 1
find $OUT/trace -type f | head -20
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
tail -5 $OUT/trace.log
head -5 $OUT/kernel_stats.csv
find $OUT/trace -type f ! -name "*stats.csv" -delete
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o run -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$C.log 2>&1 || exit 1
  f=$(find $OUT/pmc_$C -name "*counter_collection.csv" | head -1)
  grep -E "letkf_wave_kernel|Counter_Name" "$f" | head -3 > $OUT/pmc_$C.csv
  cat $OUT/pmc_$C.csv
  tail -3 $OUT/pmc_$C.log
  rm -rf $OUT/pmc_$C
done
