#!/bin/bash
OUT=$(pwd)/gpurun_out/r02; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/gpu_full.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/gpu_full.log
bash profiles/collect.sh r02 v5 2>&1 | tail -3 | cut -c1-400
