cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in base skip0 skip5 skip10; do
  if [ $lib = base ]; then unset LETKF_AMD_LIB; else export LETKF_AMD_LIB=$GRAFT_REPO_ROOT/scale-letkf_amd/lib/libletkf_amd_$lib.so; fi
  for a in "" "--ensval correlated --obs-spread 2.4"; do
  timeout -k 10 300 python3 bench.py --workload C3-slab --steps 4 --warmup 1 --no-cpu-baseline $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $a', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'iters', d.get('eigenfree_iterations_mean'), 'fallback', d.get('eigenfree_fallback_points'))"
  done
done
