#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r02; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5prof -o c5 -- python3 $ROOT/bench.py --workload rlz_1023x512x128 --storage f32 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/config5_under_rocprof.json 2> $OUT/c5prof.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/natprof -o nat -- python3 $ROOT/profiles/native_timers.py 10 > $OUT/native_under_rocprof.json 2> $OUT/natprof.log
cd $ROOT
cp $(find $OUT/c5prof -name "*kernel_stats.csv" | head -1) $OUT/config5_f32_kernel_stats.csv
cp $(find $OUT/natprof -name "*kernel_stats.csv" | head -1) $OUT/native_kernel_stats.csv
head -12 $OUT/config5_f32_kernel_stats.csv | cut -c1-70,250-400; head -9 $OUT/native_kernel_stats.csv | cut -c1-70,250-400
