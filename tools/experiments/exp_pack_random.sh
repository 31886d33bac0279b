#!/bin/bash
# random-order ingest experiments (same box): group length and non-temporal gather loads
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --workload sim1x65536_end_to_end --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs"
run() { echo -n "$1: "; shift; env "$@" $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % d['ms_per_step'], d['config'].get('regrouped_table_equals_source'))"; }
for ord in random dense; do
  B2="$B --arrival-order $ord"
  B="$B2"
  run "$ord base        " A=1
  run "$ord gt 94%      " DCARL_DP_GT_PCT=94
  run "$ord gt 90%      " DCARL_DP_GT_PCT=90
  run "$ord nt          " DCARL_HIP_LIB=tools/ab/libnt.so
  run "$ord nt + gt 94% " DCARL_HIP_LIB=tools/ab/libnt.so DCARL_DP_GT_PCT=94
  run "$ord base        " A=1
  B="python bench.py --workload sim1x65536_end_to_end --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs"
done
