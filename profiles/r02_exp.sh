#!/bin/bash
# round-2 experiment driver (GPU box): microbenchmarks, GPU test suite, bench
OUT=gpurun_out/r02; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -o /tmp/pl profiles/micro/pipeline.hip && timeout -k 10 120 /tmp/pl > $OUT/pipeline.txt 2>&1
cat $OUT/pipeline.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -s > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" $OUT/gpu_tests.log | tail -15
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>$OUT/bench.err | tee $OUT/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items()})"
