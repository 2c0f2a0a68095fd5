for rep in 1 2 3; do
for t in ref new; do
  if [ $t = new ]; then unset LETKF_AMD_LIB; else export LETKF_AMD_LIB=$PWD/scale-letkf_amd/lib/libletkf_amd_ref.so; fi
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t ms/step', round(d['ms_per_step'],1))"
done
done
LETKF_AMD_LIB=$PWD/scale-letkf_amd/lib/libletkf_amd_prof.so timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "letkf prof" | tail -1
