#!/bin/bash
OUT=$(pwd)/gpurun_out/r02; mkdir -p $OUT
timeout -k 10 300 python3 profiles/native_timers.py 20 > $OUT/native_timers.json 2>$OUT/native_timers.err; cat $OUT/native_timers.json; tail -2 $OUT/native_timers.err
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "native or config4 or ragged" > $OUT/gpu_tests_subset.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/gpu_tests_subset.log
SCYTHE_HIP_LIB=$PWD/profiles/libscythe_hip_phases.so SX_DFT_PHASES_OUT=$OUT/phases_dft.bin timeout -k 10 300 python3 profiles/native_timers.py 3 > $OUT/native_phases.json 2>$OUT/native_phases.err
python3 profiles/phases_dft.py $OUT/phases_dft.bin | tee $OUT/phases_dft.txt
