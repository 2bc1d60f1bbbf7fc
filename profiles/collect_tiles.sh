#!/bin/bash
# Kernel time ONE rank of an N-way split executes per step (all N tiles on this GPU, loopback transport; no event pairs):
#   gpurun -- 'bash profiles/collect_tiles.sh r04 8'  -> gpurun_out/prof/<round>/tiles<N>_{iface,a2a}_kernel_stats.csv + a summary line each
set -e
ROUND=${1:-r04}; N=${2:-8}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof/$ROUND; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for kind in iface a2a; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tiles_$kind -o t -- python3 $ROOT/profiles/tile_timers.py $N $kind cost notimers > $OUT/tiles${N}_$kind.log 2> $OUT/tiles${N}_$kind.err
  cp $(find $OUT/tiles_$kind -name "*kernel_stats.csv" | head -1) $OUT/tiles${N}_${kind}_kernel_stats.csv
  python3 - $OUT/tiles${N}_${kind}_kernel_stats.csv $N $kind <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sx::" in r["Name"] and "k_nan_check" not in r["Name"]]
n = int(sys.argv[2]); steps = 23      # tile_timers.py: 3 + 20 steps
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%s, %d tiles: %.1f us of kernels per rank and step (sum over kernels / (%d steps x %d tiles))" % (sys.argv[3], n, tot / 1e3 / steps / n, steps, n))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("   %-34s %8.1f us per rank and step, %6.1f us per launch" % (r["Name"].split("(")[0].replace("void ", "").replace("sx::", "")[:34], float(r["TotalDurationNs"]) / 1e3 / steps / n, float(r["AverageNs"]) / 1e3))
PY
done
