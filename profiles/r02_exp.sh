#!/bin/bash
OUT=$(pwd)/gpurun_out/r02; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -x -k "config5 or 128" > $OUT/gpu_tests_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/gpu_tests_subset.log
for seg in 0 2 3 8; do SX_SBW_SEG=$seg timeout -k 10 600 python bench.py --workload rlz_1023x512x128 --storage f32 --steps 10 --warmup 3 --no-cpu-baseline 2>$OUT/bench_c5.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config5 f32 seg $seg', round(d['value'],1), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items()})"; done
SX_SBW_MFMA=0 timeout -k 10 600 python bench.py --workload rlz_1023x512x128 --storage f32 --steps 10 --warmup 3 --no-cpu-baseline 2>$OUT/bench_c5.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config5 f32 valu', round(d['value'],1), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items()})"
