#!/bin/bash
# PROF twin: per-phase wave-time shares of the KR = 20 instantiation on C2-k20 and C1 (and KR = 50 on C2 for reference)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_prof_k20; mkdir -p $O
export LETKF_AMD_LIB=$GRAFT_REPO_ROOT/scale-letkf_amd/lib/libletkf_amd_prof.so
for w in C2-k20 C1 C2-k10; do
  timeout -k 10 300 python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $O/$w.json 2> $O/$w.err
  echo "== $w"; grep "letkf prof" $O/$w.err | tail -2; tail -1 $O/$w.json | cut -c1-300
done
