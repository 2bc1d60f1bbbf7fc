#!/bin/bash
# phase stamps of the equation-set kernel and the node FFT kernel (diagnostic build) on the GPU box
OUT=gpurun_out/r02; mkdir -p $OUT
SCYTHE_HIP_LIB=$PWD/profiles/libscythe_hip_phases.so SX_PHASES_OUT=$OUT/phases.bin SX_FFT_PHASES_OUT=$OUT/phases_fft.bin SX_SBW_PHASES_OUT=$OUT/phases_sbw.bin timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/phases_bench.json 2> $OUT/phases.err
python profiles/phases.py $OUT/phases.bin | tee $OUT/phases.txt
python profiles/phases_fft.py $OUT/phases_fft.bin | tee $OUT/phases_fft.txt
python profiles/phases_sbw.py $OUT/phases_sbw.bin | tee $OUT/phases_sbw.txt
