#!/bin/bash
# Copy the record of tools/r2_final.sh TAG from gpurun_out/ into profiles/ (the r02_* set always comes from ONE build).
TAG=${1:?tag}; G=gpurun_out; P=profiles
for f in $G/final_$TAG/bench_*.json; do b=$(basename $f); cp $f $P/r02_$b; done
cp $G/prof_$TAG/bench_c2.json $P/r02_bench_c2.json
cp $G/prof_$TAG/kernel_stats.csv $P/r02_c2_kernel_stats.csv
cp $G/prof_$TAG/pmc_FETCH_SIZE.csv $P/r02_c2_pmc_fetch.csv
cp $G/prof_$TAG/pmc_WRITE_SIZE.csv $P/r02_c2_pmc_write.csv
cp $G/sq_$TAG/sq.csv $P/r02_c2_pmc_sq.csv
for w in c3slab c4slab c5slab c2_maxnobs100; do [ -f $G/r02_${TAG}_${w}_kernel_stats.csv ] && cp $G/r02_${TAG}_${w}_kernel_stats.csv $P/r02_${w}_kernel_stats.csv; done
cp $G/final_$TAG.log $P/r02_final_record.log
