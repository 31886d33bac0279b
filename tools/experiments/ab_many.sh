#!/bin/bash
# same-box A/B of several library builds on one workload: tools/experiments/ab_many.sh "<bench args>" libX.so libY.so ...  (plus the in-tree build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
W="$1"; shift
for i in 1 2 3; do
  for v in intree "$@"; do
    if [ $v = intree ]; then unset DCARL_HIP_LIB; else export DCARL_HIP_LIB=$PWD/tools/ab/$v; fi
    python bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['roofline']['kernel'], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
  done
done
