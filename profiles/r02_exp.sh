#!/bin/bash
OUT=$(pwd)/gpurun_out/r02; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/gpu_full.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/gpu_full.log
bash profiles/collect.sh r02 v4 2>&1 | tail -4 | cut -c1-600
timeout -k 10 300 python3 profiles/native_timers.py 20 > $OUT/native_timers.json 2>/dev/null; cat $OUT/native_timers.json
SCYTHE_HIP_LIB=$PWD/profiles/libscythe_hip_phases.so SX_DFT_PHASES_OUT=$OUT/phases_dft.bin timeout -k 10 300 python3 profiles/native_timers.py 3 > /dev/null 2>&1
python3 profiles/phases_dft.py $OUT/phases_dft.bin > $OUT/phases_dft.txt
timeout -k 10 600 python bench.py --workload rlz_1023x512x128 --storage f32 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/config5_f32_bench.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/config5_f32_bench.json')); print('config5 f32', d['value'])"
timeout -k 10 600 python bench.py --workload rlz_1023x512x128 --storage f64 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/config5_f64_bench.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/config5_f64_bench.json')); print('config5 f64', d['value'])"
