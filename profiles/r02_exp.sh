#!/bin/bash
OUT=gpurun_out/r02; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "hrbl or node_space or fp32 or config4 or rz_ or rlz_advection or index_maps" > $OUT/gpu_tests_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/gpu_tests_subset.log
for v in 1 2; do
  timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-native 2>$OUT/bench_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items()})" || exit 1
done
