#!/bin/bash
OUT=gpurun_out/r02; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tiles_on_one" > $OUT/gpu_tests_subset.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/gpu_tests_subset.log
