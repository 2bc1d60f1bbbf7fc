#!/bin/bash
# SQ counter passes of the bench workload and of the native-ring run (kernel-trace only beside --pmc, as gpurun requires):
#   gpurun -- 'bash profiles/collect_sq.sh r02'   -> gpurun_out/prof/<round>/sq_*.csv, summarised by profiles/summarize_sq.py
set -e
ROUND=${1:-r02}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof/$ROUND; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-native --no-kernel-timers"
NAT="python3 $ROOT/profiles/native_timers.py 5"
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq1 -o sq1 -- $BENCH > /dev/null 2> $OUT/sq1.log
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq2 -o sq2 -- $BENCH > /dev/null 2> $OUT/sq2.log
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/sqn -o sqn -- $NAT > /dev/null 2> $OUT/sqn.log
cd $ROOT
for t in sq1 sq2 sqn; do cp $(find $OUT/$t -name "*counter_collection.csv" | head -1) $OUT/${t}_counter_collection.csv; done
python3 profiles/summarize_sq.py $OUT/sq1_counter_collection.csv $OUT/sq2_counter_collection.csv > $OUT/sq_summary_bench.txt
python3 profiles/summarize_sq.py $OUT/sqn_counter_collection.csv > $OUT/sq_summary_native.txt
cat $OUT/sq_summary_bench.txt $OUT/sq_summary_native.txt
