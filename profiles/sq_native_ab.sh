#!/bin/bash
# SQ counters of the native-ring inverse DFT kernel under settings of one environment variable (kernel-trace only beside --pmc):
#   gpurun -- 'bash profiles/sq_native_ab.sh SX_DFT_EIGHTH 0 2'
VAR=$1; shift
ROOT=$PWD; OUT=$ROOT/gpurun_out/r04/sq_native; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for val in "$@"; do
  export $VAR=$val
  timeout -k 10 250 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/${VAR}_$val -o sq -- python3 $ROOT/profiles/native_timers.py 5 85 > $OUT/${VAR}_$val.json 2> $OUT/${VAR}_$val.log
  f=$(find $OUT/${VAR}_$val -name "*counter_collection.csv" | sort | tail -1)
  echo "== $VAR=$val"
  if [ -n "$f" ]; then python3 $ROOT/profiles/summarize_sq.py "$f" > $OUT/${VAR}_$val.txt; grep -E "^kernel|k_rl_inverse" $OUT/${VAR}_$val.txt; fi
done
