#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the headline kernel only (two PMC passes), for a library variant: tools/r2_pmc_quick.sh TAG [variant]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-q}; V=${2:-}
[ -n "$V" ] && export LETKF_AMD_LIB=$GRAFT_REPO_ROOT/scale-letkf_amd/lib/libletkf_amd_$V.so
OUT=gpurun_out/pmcq_$TAG; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o run -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$C.log 2>&1 || { tail -5 $OUT/pmc_$C.log; exit 1; }
  f=$(find $OUT/pmc_$C -name "*counter_collection.csv" | head -1)
  grep -E "letkf_wave_kernel" "$f" | head -1 | awk -F, '{print $(NF-3), $(NF-2), "scratch", $12}'
  rm -rf $OUT/pmc_$C
done
