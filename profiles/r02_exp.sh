#!/bin/bash
OUT=gpurun_out/r02; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "fft_rings or 512 or config5 or fp32 or rl_slab or rl_advection or node_space or config4" > $OUT/gpu_tests_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/gpu_tests_subset.log
timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-native 2>$OUT/bench_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', round(d['value'],1), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items()})"
timeout -k 10 600 python bench.py --workload rlz_1023x512x128 --storage f32 --steps 10 --warmup 3 --no-cpu-baseline 2>$OUT/bench_c5.err | tee $OUT/bench_c5.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config5 f32', round(d['value'],1), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items()})"
