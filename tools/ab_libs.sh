#!/bin/bash
# same-box comparison of library builds ("cur" = the in-tree build):
#   gpurun -- 'bash tools/ab_libs.sh "--workload sim2_ragged_batch" tools/ab/libX.so cur'
args=$1; shift
for rep in 1 2; do
for l in "$@"; do
  if [ "$l" = cur ]; then unset DCARL_HIP_LIB; else export DCARL_HIP_LIB=$PWD/$l; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l', round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))"
done
done
