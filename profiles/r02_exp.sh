#!/bin/bash
OUT=gpurun_out/r02; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -s -k "config5 or fp32_storage_512" --durations=8 > $OUT/gpu_tests_c5.log 2>&1; echo "pytest rc=$?"; grep -E "config 5|fp32-stored|passed|failed|Error|error|s call" $OUT/gpu_tests_c5.log | tail -20
