#!/bin/bash
# round-2 experiment driver (GPU box): microbenchmark, A/B of the 16-byte-per-lane paths, GPU test suite
OUT=gpurun_out/r02; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -o /tmp/lw profiles/micro/lane_width.hip && timeout -k 10 120 /tmp/lw > $OUT/lane_width.txt 2>&1
cat $OUT/lane_width.txt
for v in 0 1 0 1; do
  SX_WIDE=$v timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>$OUT/bench_wide$v.err | tee $OUT/bench_wide$v.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SX_WIDE=$v', round(d['value'],1), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items()})" || exit 1
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/gpu_tests.log
