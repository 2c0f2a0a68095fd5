#!/bin/bash
# Builds the library of another commit as scale-letkf_amd/lib/libletkf_amd_ref.so (git worktree under /tmp), for
# same-box A/B runs on the GPU:  LETKF_AMD_LIB=$PWD/scale-letkf_amd/lib/libletkf_amd_ref.so python bench.py ...
# Box-to-box spread on the pool is 1-3 % (up to 10 % on the large-k paths); an A/B has to run both builds in one gpurun call.
# Usage: tools/ab_build_ref.sh [commit]   (default HEAD)
set -e
C=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WT=/tmp/letkf_ab_ref
git -C "$ROOT" worktree remove --force $WT 2>/dev/null || true
git -C "$ROOT" worktree add -f $WT "$C" -q
make -C $WT/scale-letkf_amd -j6 OUT=$ROOT/scale-letkf_amd/lib/libletkf_amd_ref.so OBJDIR=$WT/obj > /tmp/letkf_ab_ref.log 2>&1 || { tail -5 /tmp/letkf_ab_ref.log; exit 1; }
git -C "$ROOT" worktree remove --force $WT
ls -la $ROOT/scale-letkf_amd/lib/libletkf_amd_ref.so
