#!/bin/bash
OUT=gpurun_out/r02; mkdir -p $OUT
for v in "SX_X=0" "SX_CELL128_LAM4=1"; do
env $v timeout -k 10 600 python bench.py --workload rlz_1023x512x128 --storage f32 --steps 10 --warmup 3 --no-cpu-baseline 2>$OUT/bench_c5.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config5 f32 $v', round(d['value'],1), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items()})"
done
SX_CELL128_LAM4=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "128" 2>&1 | tail -2
