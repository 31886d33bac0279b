#!/bin/bash
# same-box A/B of tools/ab/lib<V>.so builds (tools/build_trace_variants.sh) on the three online workloads:
#   VARIANTS="B P0 S0" bash tools/experiments/ab_libs_online.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
  for v in ${VARIANTS:-B}; do
    export DCARL_HIP_LIB=$PWD/tools/ab/lib$v.so
    for w in "sim1x65536_trace" "cfg3_sim2_argmax --mode trace" "cfg4_mixed --total-states 524288 --mode trace"; do
      python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$w'[:24], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
    done
  done
done
