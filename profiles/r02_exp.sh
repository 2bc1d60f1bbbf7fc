#!/bin/bash
OUT=$(pwd)/gpurun_out/r02; mkdir -p $OUT
( time timeout -k 10 800 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -s -k "native_equivalent_full_size" ) > $OUT/gpu_tests_subset.log 2>&1; echo "pytest rc=$?"; tail -16 $OUT/gpu_tests_subset.log
