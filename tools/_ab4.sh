for rep in 1 2; do
for t in ref new; do
  if [ $t = new ]; then unset LETKF_AMD_LIB; else export LETKF_AMD_LIB=$PWD/scale-letkf_amd/lib/libletkf_amd_ref.so; fi
  echo "== $t"
  timeout -k 10 300 python bench_largek.py 2>&1 | grep -E "^(64|100) " | cut -c1-125
  for w in C2-mini-k100 C2-mini-k20 C2-mini; do timeout -k 10 200 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w ms/step', round(d['ms_per_step'],3), int(d['value']))"; done
done
done
