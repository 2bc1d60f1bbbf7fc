#!/bin/bash
# Collect the round's profiles on the GPU box:  gpurun -- 'bash profiles/collect.sh r01 v7'
# 1) rocprofv3 --kernel-trace --stats of bench.py; 2)+3) separate --pmc FETCH_SIZE / WRITE_SIZE passes (kernel-trace
# only, as MI355X_MICROARCH.md prescribes); 4) the un-profiled default bench line.  Results land under gpurun_out/prof/
# and are copied from there into profiles/<round>/ by hand (see profiles/README.md).
set -e
ROUND=${1:-r01}; TAG=${2:-v7}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof/$ROUND; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- $BENCH > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $BENCH > /dev/null 2> $OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $BENCH > /dev/null 2> $OUT/write.log
cd $ROOT
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.log
find $OUT -name "*.csv" | head -20
tail -c 1500 $OUT/${TAG}_bench.json
