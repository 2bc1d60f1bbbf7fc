#!/bin/bash
OUT=$(pwd)/gpurun_out/r02; mkdir -p $OUT
( time timeout -k 10 1000 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -s -k "config5_full_size_state_against" ) > $OUT/gpu_tests_subset.log 2>&1; echo "pytest rc=$?"; tail -14 $OUT/gpu_tests_subset.log
