#!/bin/bash
# same-box comparison of the trace kernel mappings:  gpurun -- 'bash tools/experiments/ab_kernels.sh single tab ...'
for rep in 1 2; do
for k in "$@"; do
  DCARL_TRACE_KERNEL=$k python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$k', round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))"
done
done
