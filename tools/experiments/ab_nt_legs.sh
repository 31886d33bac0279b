#!/bin/bash
# same-box A/B of two builds (A = tools/ab/libA.so, N = tools/ab/libN.so) on the legs whose kernels the non-temporal variants touch
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V="${VARIANTS:-A N A N}"
for v in $V; do
  export DCARL_HIP_LIB=$PWD/tools/ab/lib$v.so
  for w in "sampler_pairs --records 1073741824" "sim1x65536_batch" "cfg3_sim2_argmax" "cfg4_mixed --total-states 524288" "sim1x65536_end_to_end" "sim1x65536_end_to_end --arrival-order random"; do
    python bench.py --workload $w --steps 12 --warmup 4 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$w'[:44].ljust(44), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
  done
done
