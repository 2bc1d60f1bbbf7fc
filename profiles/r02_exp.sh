#!/bin/bash
OUT=gpurun_out/r02; mkdir -p $OUT
for v in "SX_HIST_EARLY=0" "SX_HIST_EARLY=1" "SX_HIST_EARLY=0" "SX_HIST_EARLY=1"; do
  env $v timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-native 2>$OUT/bench_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['kernels_ms_per_step']['k_phys_hrbl'],4), d['config']['nan'])" || exit 1
done
SX_HIST_EARLY=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hrbl or node_space" 2>&1 | tail -2
