#!/bin/bash
OUT=$(pwd)/gpurun_out/r02; mkdir -p $OUT
for w in 1 0 1024; do SX_WALK=$w timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-native 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('walk $w', round(d['value'],1), d['dominant_kernel_ms_timed_region'], {k: round(x,4) for k,x in d['kernels_ms_per_step'].items()})"; done
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hrbl or node" > $OUT/gpu_tests_subset.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/gpu_tests_subset.log
