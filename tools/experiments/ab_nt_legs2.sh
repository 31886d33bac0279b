#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in A N1 N2 A N1 N2; do
  export DCARL_HIP_LIB=$PWD/tools/ab/lib$v.so
  for w in "sim1x65536_end_to_end" "sim1x65536_end_to_end --arrival-order random" "sampler_to_estimator"; do
    python bench.py --workload $w --steps 12 --warmup 4 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$w'[:44].ljust(44), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
  done
done
