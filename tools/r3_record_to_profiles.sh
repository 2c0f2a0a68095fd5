#!/bin/bash
# Copy the record of tools/r3_final.sh TAG (parts A and B) from gpurun_out/ into profiles/ (the r03_* set comes from ONE build).
TAG=${1:?tag}; G=gpurun_out; P=profiles
for f in $G/final_$TAG/bench_*.json; do b=$(basename $f); cp $f $P/r03_$b; done
cp $G/prof_$TAG/bench_c2.json $P/r03_bench_c2.json
cp $G/prof_$TAG/kernel_stats.csv $P/r03_c2_kernel_stats.csv
cp $G/prof_$TAG/pmc_FETCH_SIZE.csv $P/r03_c2_pmc_fetch.csv
cp $G/prof_$TAG/pmc_WRITE_SIZE.csv $P/r03_c2_pmc_write.csv
cp $G/sq_$TAG/sq.csv $P/r03_c2_pmc_sq.csv
for f in $G/spread_$TAG/bench_*.json; do b=$(basename $f); cp $f $P/r03_spread_$b; done
for t in c3slab c3slab24 k100 c5slab c2k20; do d=$G/pmc_${TAG}_$t; [ -d $d ] || continue
  cp $d/summary.json $P/r03_${t}_pmc_summary.json; cp $d/kernel_stats.csv $P/r03_${t}_kernel_stats.csv
  cp $d/pmc_FETCH_SIZE.csv $P/r03_${t}_pmc_fetch.csv; cp $d/pmc_WRITE_SIZE.csv $P/r03_${t}_pmc_write.csv; [ -f $d/pmc_sq.csv ] && cp $d/pmc_sq.csv $P/r03_${t}_pmc_sq.csv; done
for f in $G/full_$TAG/bench_*.json; do b=$(basename $f .json); cp $f $P/r03_${b}_full.json; done
cat $G/final_${TAG}_A.log $G/final_${TAG}_B.log > $P/r03_final_record.log 2>/dev/null
