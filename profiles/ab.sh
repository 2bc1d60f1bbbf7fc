#!/bin/bash
# A/B timing of two builds of libscythe_hip.so on ONE GPU box (box-to-box variance is 1-2 %, run-to-run on a box ~0.2 %):
#   cp scythe.jl_amd/libscythe_hip.so profiles/lib_A.so; <change, make>; cp scythe.jl_amd/libscythe_hip.so profiles/lib_B.so
#   gpurun -- 'bash profiles/ab.sh'
# (the .so files are git-ignored; SCYTHE_HIP_LIB makes scythe.jl_amd/_lib.py load the given build)
for r in 1 2; do for v in A B; do
  SCYTHE_HIP_LIB=$PWD/profiles/lib_$v.so python bench.py --steps 150 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value'],1), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items()})" $v || exit 1
done; done
