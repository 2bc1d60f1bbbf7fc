#!/bin/bash
# A/B/A/B of ONE library build under two settings of an environment switch read at sx_create, on one GPU box:
#   gpurun -- 'bash profiles/ab_env.sh SX_FFT_REG 0 1'
VAR=$1; A=$2; B=$3
for r in 1 2; do for v in $A $B; do
  env $VAR=$v python bench.py --steps 100 --warmup 100 --no-cpu-baseline --no-native --no-other-configs --schedule serial 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value'],1), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items()})" "$VAR=$v" || exit 1
done; done
