# Parity margins (tools/parity_margin.py) and C2 speed for library builds with different early-stop constants, e.g.
#   make -C scale-letkf_amd OBJDIR=$PWD/scale-letkf_amd/lib/obj_x OUT=$PWD/scale-letkf_amd/lib/libletkf_amd_x.so \
#        CXXFLAGS="-O3 -std=c++17 -fPIC -DLETKF_EARLY_TOL2=1e-14 -DLETKF_EARLY_T2=1e-10"
# Usage (GPU box): bash tools/run_tol.sh default x ...
for t in "$@"; do
  if [ $t = default ]; then unset LETKF_AMD_LIB; else export LETKF_AMD_LIB=$PWD/scale-letkf_amd/lib/libletkf_amd_$t.so; fi
  echo "== build $t"
  timeout -k 10 300 python tools/parity_margin.py 2>&1 | tail -31 || exit 1
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 ms/step', round(d['ms_per_step'],1), 'sweeps', d.get('jacobi_sweeps_mean'), 'bad', d.get('nonzero_status_points'))" || exit 1
done
