#!/bin/bash
# phase stamps of the node FFT kernel with the passes through LDS (SX_FFT_REG=0) and in registers (=1); diagnostic build
OUT=gpurun_out/r04; mkdir -p $OUT
for v in 0 1; do
  SX_FFT_REG=$v SCYTHE_HIP_LIB=$PWD/profiles/libscythe_hip_phases.so SX_FFT_PHASES_OUT=$OUT/phases_fft_$v.bin timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-native --no-other-configs --schedule serial > /dev/null 2> $OUT/phases_$v.err
  echo "SX_FFT_REG=$v"; python profiles/phases_fft.py $OUT/phases_fft_$v.bin
done | tee $OUT/phases_fft_reg_ab.txt
