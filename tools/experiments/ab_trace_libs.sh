#!/bin/bash
# same-box A/B of builds of the online kernel: VARIANTS="A P2 P6" bash tools/experiments/ab_trace_libs.sh   (tools/ab/lib<V>.so via DCARL_HIP_LIB)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
  for v in ${VARIANTS:-A}; do
    export DCARL_HIP_LIB=$PWD/tools/ab/lib$v.so
    for w in "sim1x65536_trace" "cfg3_sim2_argmax --mode trace" "cfg4_mixed --total-states 524288 --mode trace"; do
      python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$w'[:24], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
    done
  done
done
