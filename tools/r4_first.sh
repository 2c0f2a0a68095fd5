#!/bin/bash
# round 4, first GPU call: the FP64 pipe-overlap micro-benchmark, the new oracle-backed tests, the whole GPU suite, one bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_first; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_fp64_overlap.hip -o $O/ubench_fp64_overlap 2>/dev/null && timeout -k 10 120 $O/ubench_fp64_overlap > $O/ubench_fp64_overlap.log 2>&1; cat $O/ubench_fp64_overlap.log
echo "== new tests"; timeout -k 10 1100 python3 -m pytest tests/test_gpu_columns.py tests/test_gpu_checked.py tests/test_gpu_configs.py -m gpu -q -x > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
