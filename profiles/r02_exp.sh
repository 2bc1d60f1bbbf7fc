#!/bin/bash
OUT=gpurun_out/r02; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "native or rlz_hrbl or fp32_storage_mode or rlz_advection" > $OUT/gpu_tests_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/gpu_tests_subset.log
timeout -k 10 300 python profiles/time_configs.py native4 2>&1 | tail -1
