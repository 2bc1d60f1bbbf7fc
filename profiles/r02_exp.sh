#!/bin/bash
OUT=$(pwd)/gpurun_out/r02; mkdir -p $OUT
for r in 0 1 8; do SX_ZINV_RPW=$r timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-native 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rpw $r', round(d['value'],1), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items() if k in ('k_zinv','k_node_fft','k_rl_inverse')})"; done
timeout -k 10 600 python bench.py --workload rlz_1023x512x128 --storage f32 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config5 f32', round(d['value'],1), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items()})"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "rz or rlz or hrbl or config3 or config5 or tiles" > $OUT/gpu_tests_subset.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/gpu_tests_subset.log
