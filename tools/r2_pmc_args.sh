#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the loop-body kernel for a bench.py run with arbitrary arguments (separate passes):
# tools/r2_pmc_args.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null > $OUT/bench.json; tail -c 1500 $OUT/bench.json | head -c 700; echo
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $OUT/p_$C -o run -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-search-in-step "$@" > $OUT/pmc_$C.log 2>&1 || { tail -5 $OUT/pmc_$C.log; exit 1; }
  f=$(find $OUT/p_$C -name "*counter_collection.csv" | head -1)
  grep -E "letkf_wave_kernel|letkf_eig|letkf_stage|Counter_Name" "$f" | head -4 > $OUT/pmc_$C.csv
  cut -d, -f9,16,17 $OUT/pmc_$C.csv
  rm -rf $OUT/p_$C
done
