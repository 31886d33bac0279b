#!/bin/bash
# Same-box A/B of two builds of libdcarl_hip.so on any bench workload (A = tools/ab/libA.so via DCARL_HIP_LIB, B = in-tree):
#   cp dcarl_amd/libdcarl_hip.so tools/ab/libA.so; <edit>; python dcarl_amd/build.py; gpurun -- 'bash tools/experiments/ab_libs.sh "cfg3_sim2_argmax" "cfg4_mixed --total-states 524288"'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2 3; do
  for w in "$@"; do
    for v in A B; do
      if [ $v = A ]; then export DCARL_HIP_LIB=$PWD/tools/ab/libA.so; else unset DCARL_HIP_LIB; fi
      python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$w', d['roofline']['kernel'], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
    done
  done
done
