#!/bin/bash
# rocprofv3 kernel statistics of the native-ring run under two settings of one environment variable:
#   gpurun -- 'bash profiles/ab_native_rocprof.sh SX_DFT_EIGHTH 0 2'
VAR=$1; shift
ROOT=$PWD; OUT=$ROOT/gpurun_out/r04/ab_native; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for val in "$@"; do
  export $VAR=$val
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${VAR}_$val -o run -- python3 $ROOT/profiles/native_timers.py 10 85 > $OUT/${VAR}_$val.json 2> $OUT/${VAR}_$val.log
  f=$(find $OUT/${VAR}_$val -name "*kernel_stats.csv" | sort | tail -1)
  echo "== $VAR=$val  $(cat $OUT/${VAR}_$val.json | cut -c1-120)"
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 2.0:
        print("  %-60s calls %5s avg %9.1f us  min %9.1f max %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  fi
done
