#!/bin/bash
# Collect a round's profiles on the GPU box:  gpurun -- 'bash profiles/collect.sh r03 v1'
# 1) rocprofv3 --kernel-trace --stats of bench.py; 2)+3) separate --pmc FETCH_SIZE / WRITE_SIZE passes (kernel-trace
# only, as MI355X_MICROARCH.md prescribes); 4) per-kernel HBM bytes (summarize_pmc.py, with the library's sha256);
# 5) the un-profiled default bench line.  Results land under gpurun_out/prof/<round>/ and are copied from there into
# profiles/<round>/ (see profiles/README.md).
set -e
ROUND=${1:-r04}; TAG=${2:-v1}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof/$ROUND; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-native --no-other-configs --schedule serial"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- $BENCH > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $BENCH > /dev/null 2> $OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $BENCH > /dev/null 2> $OUT/write.log
cd $ROOT
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
python3 profiles/summarize_pmc.py $F $W $OUT/pmc_traffic_rlz_513x256x64.json scythe.jl_amd/libscythe_hip.so > $OUT/pmc_summary.txt
cp $F $OUT/fetch_size_counter_collection.csv; cp $W $OUT/write_size_counter_collection.csv
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.log
cat $OUT/pmc_summary.txt
head -12 $OUT/${TAG}_kernel_stats.csv
tail -c 2500 $OUT/${TAG}_bench.json
