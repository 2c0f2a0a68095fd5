cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
j() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'solves/s', int(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'iters', round(d['eigenfree_iterations_mean'],2), 'parity', d.get('parity_sample_max_rel'))"; }
for w in "C3-slab" "C3-slab --ensval correlated --obs-spread 2.4" "C2-slab-k100 --ensval correlated --obs-spread 2.4" "C5-slab --ensval correlated --obs-spread 4.0"; do
  for lib in base t28 t26; do
    if [ $lib = base ]; then unset LETKF_AMD_LIB; else export LETKF_AMD_LIB=$GRAFT_REPO_ROOT/scale-letkf_amd/lib/libletkf_amd_$lib.so; fi
    timeout -k 10 300 python3 bench.py --workload $w --steps 4 --warmup 1 --cpu-seconds 3 2>/dev/null | j "$lib $w"
  done
done
