#!/bin/bash
# same-box comparison of library builds:  gpurun -- 'DCARL_TRACE_KERNEL=tab bash tools/ab_libs.sh tools/ab/libX.so tools/ab/libY.so'
for rep in 1 2; do
for l in "$@"; do
  DCARL_HIP_LIB=$PWD/$l python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l', round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))"
done
done
