#!/bin/bash
# SQ occupancy/issue counters of the headline kernel (one PMC pass, its own run).  Usage: ./tools/pmc_sq.sh TAG
set -o pipefail
TAG=${1:-vX}
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/pmc -o run -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc.log 2>&1 || { tail -20 $OUT/pmc.log; exit 1; }
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
grep -E "letkf_wave_kernel|Counter_Name" "$f" > $OUT/sq.csv
python - "$OUT/sq.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    print(r["Counter_Name"], r["Counter_Value"])
PY
tail -2 $OUT/pmc.log
rm -rf $OUT/pmc
