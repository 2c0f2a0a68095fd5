#!/bin/bash
# Round 3: counters of the staged path's kernels: kernel stats, FETCH_SIZE, WRITE_SIZE and an SQ pass,
# each in its own run, for one bench.py workload.  Usage: tools/r3_pmc.sh TAG <bench args...>  -> gpurun_out/pmc_TAG/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --cpu-seconds 4 "$@" > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; ; rm -rf $O/trace
grep -E "letkf::|\"Name\"" $O/kernel_stats.csv | cut -c1-200
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $O/p_$C -o run -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $O/pmc_$C.log 2>&1 || { tail -5 $O/pmc_$C.log; exit 1; }
  f=$(find $O/p_$C -name "*counter_collection.csv" | head -1)
  grep -E "letkf::|Counter_Name" "$f" > $O/pmc_$C.csv
  rm -rf $O/p_$C
done
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE \
  --output-format csv -d $O/p_sq -o run -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $O/pmc_sq.log 2>&1 || { tail -5 $O/pmc_sq.log; }
f=$(find $O/p_sq -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && grep -E "letkf::|Counter_Name" "$f" > $O/pmc_sq.csv
rm -rf $O/p_sq
python3 tools/r3_pmc_summary.py $O
