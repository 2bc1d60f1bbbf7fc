#!/bin/bash
OUT=$(pwd)/gpurun_out/r02; mkdir -p $OUT
for i in 1 2; do timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-native 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', round(d['value'],1), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items()})"; done
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fft or node or tiles or slab or hrbl" > $OUT/gpu_tests_subset.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/gpu_tests_subset.log
