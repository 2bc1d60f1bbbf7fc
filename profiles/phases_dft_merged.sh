#!/bin/bash
OUT=gpurun_out/r04; mkdir -p $OUT
SCYTHE_HIP_LIB=$PWD/profiles/libscythe_hip_phases.so SX_DFT_PHASES_OUT=$OUT/phases_dft_merged.bin timeout -k 10 300 python profiles/native_timers.py 3 > /dev/null 2> $OUT/phases_dft.err
python profiles/phases_dft_merged.py $OUT/phases_dft_merged.bin | tee $OUT/phases_dft_merged.txt
