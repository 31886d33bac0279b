#!/bin/bash
# sampler stores: plain (tools/ab/libS0.so, -DDCARL_SAMPLER_NT=0) against non-temporal (in-tree), alternating on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2 3 4; do
  for v in S0 B; do
    if [ $v = B ]; then unset DCARL_HIP_LIB; else export DCARL_HIP_LIB=$PWD/tools/ab/lib$v.so; fi
    for w in "sampler_pairs --records 1073741824" "sampler_pairs --records 268435456"; do
      python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$w'[:40].ljust(40), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
    done
  done
done
