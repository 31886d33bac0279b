#!/bin/bash
# the final-state entries of the DEFAULT bench run (other_configs) with the launcher's own instance choice vs DCARL_QUAD=4,4,2
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
for r in 1 2 3; do
  for v in auto 4,4,2; do
    if [ $v = auto ]; then unset DCARL_QUAD; else export DCARL_QUAD=$v; fi
    python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d['other_configs']
print('$v', {k:(o[k]['kernel'], round(o[k]['kernel_ms'],4)) for k in ('configs[3].batch','configs[4].batch','configs[2].1e6_pairs')})"
  done
done
