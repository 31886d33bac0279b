#!/bin/bash
# same-box A/B of tools/ab/lib<V>.so builds on the three final-state workloads:
#   VARIANTS="BO B0 B1" bash tools/experiments/ab_libs_batch.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
  for v in ${VARIANTS:-BO}; do
    export DCARL_HIP_LIB=$PWD/tools/ab/lib$v.so
    for w in "sim1x65536_batch" "cfg3_sim2_argmax" "cfg4_mixed --total-states 524288"; do
      python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$w'[:24], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
    done
  done
done
