#!/bin/bash
# rocprofv3 kernel statistics of BASELINE.json configs 2 and 3 (bench.py --workload ...):  gpurun -- 'bash profiles/collect_configs.sh r04'
# -> gpurun_out/prof/<round>/config{2,3}_kernel_stats.csv (+ the bench line under the profiler); copy into profiles/<round>/.
set -e
ROUND=${1:-r04}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof/$ROUND; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for pair in "config2 rl_cha_bell2024" "config3 rz_513x128_semi"; do
  set -- $pair
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$1 -o $1 -- python3 $ROOT/bench.py --workload $2 --steps 200 --warmup 20 --no-graph-replay ${EXTRA_BENCH_ARGS} > $OUT/$1_bench_under_rocprof.json 2> $OUT/stats_$1.log
  cp $(find $OUT/stats_$1 -name "*kernel_stats.csv" | head -1) $OUT/$1_kernel_stats.csv
  head -12 $OUT/$1_kernel_stats.csv
done
