#!/bin/bash
# HBM traffic (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate passes) of the bench legs tools/profile_round.sh does not cover, so that
# no leg of the driver-run line carries `"traffic": null` (VERDICT r4 item 3 / What's weak 9):
#   gpurun -- 'bash tools/pmc_legs.sh r05'   ->  gpurun_out/legs_<tag>/  (+ summary: <tag>_pmc_legs.csv, hbm_traffic.json merged)
# Every pass runs under its own timeout; counters only with --kernel-trace (no other trace domain).
set -u
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/legs_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
while IFS='|' read -r NAME ARGS; do
  [ -z "$NAME" ] && continue
  B="python bench.py $ARGS --no-cpu-baseline --no-other-configs"
  timeout -k 5 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/${NAME}_fetch" -o p --output-format csv -- $B > "$OUT/$NAME.json" 2> "$OUT/${NAME}_fetch.err" || echo "$NAME fetch pass: rc $?"
  timeout -k 5 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/${NAME}_write" -o p --output-format csv -- $B > /dev/null 2> "$OUT/${NAME}_write.err" || echo "$NAME write pass: rc $?"
done <<'LEGS'
e2e_random|--workload sim1x65536_end_to_end --arrival-order random --steps 3 --warmup 1 --no-check
e2e_dense|--workload sim1x65536_end_to_end --steps 3 --warmup 1 --no-check
bft|--workload sim1x65536_batch_from_table --steps 3 --warmup 1 --no-check
buckets|--workload sim1x65536_buckets_from_table --steps 3 --warmup 1 --no-check
final_table|--workload sim1x65536_final_table --steps 5 --warmup 1
pairs_1e6|--workload sampler_pairs --steps 5 --warmup 1
pairs_2p30|--workload sampler_pairs --records 1073741824 --steps 5 --warmup 1
s2e|--workload sampler_to_estimator --records 268435456 --steps 3 --warmup 1 --no-check
s2l|--workload sampler_into_layout --records 268435456 --steps 3 --warmup 1
host_streamed|--workload sim1x65536_host_streamed --records 4096 --steps 2 --no-check
LEGS
python tools/leg_traffic.py "$OUT" "$TAG"
