#!/bin/bash
# Same-box A/B of two builds of libdcarl_hip.so (box-to-box variance on the pool is +-5-10 %):
#   1. build the baseline revision:   git stash; python dcarl_amd/build.py --force; cp dcarl_amd/libdcarl_hip.so tools/ab/libA.so; git stash pop
#   2. build the candidate:           python dcarl_amd/build.py --force
#   3. gpurun -- 'bash tools/experiments/ab_bench.sh [bench.py args]'
for i in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then export DCARL_HIP_LIB=$PWD/tools/ab/libA.so; else unset DCARL_HIP_LIB; fi
    python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))"
  done
done
