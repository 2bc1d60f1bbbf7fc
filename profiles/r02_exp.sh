#!/bin/bash
OUT=$(pwd)/gpurun_out/r02; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "native" > $OUT/gpu_tests_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/gpu_tests_subset.log
timeout -k 10 300 python3 profiles/native_timers.py 20 > $OUT/native_timers.json 2>$OUT/native_timers.err; cat $OUT/native_timers.json
SCYTHE_HIP_LIB=$PWD/profiles/libscythe_hip_phases.so SX_DFT_PHASES_OUT=$OUT/phases_dft.bin timeout -k 10 300 python3 profiles/native_timers.py 3 > $OUT/native_phases.json 2>$OUT/native_phases.err; cat $OUT/native_phases.json
python3 profiles/phases_dft.py $OUT/phases_dft.bin | tee $OUT/phases_dft.txt
