#!/bin/bash
OUT=$(pwd)/gpurun_out/r02; mkdir -p $OUT
for w in 0 1 0 1; do SX_SBW_T256=$w timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-native 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('t256 $w', round(d['value'],1), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items() if k=='k_sbz'})"; done
for sg in 8 12 16; do SX_SBW_SEG=$sg SX_SBW_T256=1 timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-native 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('t256 seg $sg', round(d['value'],1), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items() if k=='k_sbz'})"; done
SX_SBW_T256=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hrbl or node or tiles" > $OUT/gpu_tests_subset.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/gpu_tests_subset.log
