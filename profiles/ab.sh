for r in 1 2; do for v in A B; do
  SCYTHE_HIP_LIB=$PWD/profiles/lib_$v.so python bench.py --steps 150 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value'],1), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items()})" $v || exit 1
done; done
