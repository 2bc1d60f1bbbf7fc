#!/bin/bash
OUT=$(pwd)/gpurun_out/r02; mkdir -p $OUT
for ov in 0 1 0 1; do SX_OVERLAP=$ov timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-native --no-kernel-timers 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('overlap $ov', round(d['value'],1), d['ms_per_step'])"; done
