#!/bin/bash
# the pack's item order (buckets per XCD set) on both arrival orders, same box:  gpurun -- 'bash tools/experiments/exp_dp_set.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo -n "$1: "; shift; env "$@" python bench.py --workload sim1x65536_end_to_end --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --arrival-order $ORD 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % d['ms_per_step'], d['config'].get('regrouped_table_equals_source'))"; }
for ORD in random dense; do
  run "$ORD set 4 (shipped)" A=1
  run "$ORD set 2          " DCARL_HIP_LIB=tools/ab/libset2.so
  run "$ORD set 1          " DCARL_HIP_LIB=tools/ab/libset1.so
  run "$ORD set 4 (shipped)" A=1
done
