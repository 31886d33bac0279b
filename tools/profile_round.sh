#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01
# 1) --kernel-trace --stats of the default bench command, 2) separate --pmc passes for HBM traffic
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950: TCC has 4 slots, they cost 3 + 2).
# Every rocprofv3 pass runs under its own timeout: one that aborts on an uncollectable counter group does not exit by itself.
set -u
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs"
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench --output-format csv -- $BENCH > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/fetch.err"
timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/write.err"
timeout -k 5 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d "$OUT/sq" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/sq.err"
python bench.py --steps 10 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"
# the other kernels of the path (one JSON line each; not the headline): final-state CSR / ragged / dense, sampler
: > "$OUT/other_workloads.jsonl"
for W in "sim1x65536_batch" "cfg3_sim2_argmax" "cfg3_sim2_argmax --mode trace" "cfg4_mixed --total-states 524288" \
         "cfg4_mixed --total-states 524288 --mode trace" "dropin_a30_f64" "sampler_pairs" "rls_field" "frenet_candidates" "frenet_plan" "episodes" "state_ids" \
         "sim1x65536_end_to_end" "sim1x65536_batch_from_table" "sampler_to_estimator"; do
  python bench.py --workload $W --steps 10 --warmup 2 >> "$OUT/other_workloads.jsonl" 2>> "$OUT/bench.err"
done
python bench.py --workload sampler_pairs --records 1073741824 --steps 10 --warmup 2 >> "$OUT/other_workloads.jsonl" 2>> "$OUT/bench.err"
# the final-state kernel on the configs[3] / configs[4] shapes: kernel stats + HBM traffic + SQ counters (separate passes)
for C in "cfg3_sim2_argmax" "cfg4_mixed --total-states 524288"; do
  T=$(echo $C | cut -d_ -f1)
  B="python bench.py --workload $C --steps 5 --warmup 1 --no-cpu-baseline"
  timeout -k 5 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_$T" -o bench --output-format csv -- $B > /dev/null 2>> "$OUT/stats.err"
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 5 400 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/pmc_${T}_g$i" -o bench --output-format csv -- $B > /dev/null 2>> "$OUT/stats.err"
  done
done
# HBM traffic of the remaining driver-timed shapes (FETCH_SIZE and WRITE_SIZE passes only): tag|bench arguments
while IFS='|' read -r T C; do
  B="python bench.py --workload $C --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs"
  timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_${T}_g1" -o bench --output-format csv -- $B > /dev/null 2>> "$OUT/stats.err"
  timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_${T}_g2" -o bench --output-format csv -- $B > /dev/null 2>> "$OUT/stats.err"
done <<'SHAPES'
c1batch|sim1x65536_batch
pairs|sampler_pairs --records 1073741824
c3trace|cfg3_sim2_argmax --mode trace
c4trace|cfg4_mixed --total-states 524288 --mode trace
dropin|dropin_a30_f64
SHAPES
# SQ / LDS counters of the online kernel (bank conflicts of the count-root table reads)
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
i=0
for grp in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 5 400 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/pmc_trace_g$i" -o bench --output-format csv -- $B > /dev/null 2>> "$OUT/stats.err"
done
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_batch" -o bench --output-format csv -- python bench.py --workload sim1x65536_batch --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>> "$OUT/stats.err"
# round 3: the chain from the arrival-ordered (N,4) f64 table (ingest kernels + the online kernel): kernel stats + HBM traffic
B="python bench.py --workload sim1x65536_end_to_end --steps 3 --warmup 1"
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_e2e" -o bench --output-format csv -- $B > "$OUT/bench_e2e.json" 2>> "$OUT/stats.err"
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_e2e_g1" -o bench --output-format csv -- $B > /dev/null 2>> "$OUT/stats.err"
timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_e2e_g2" -o bench --output-format csv -- $B > /dev/null 2>> "$OUT/stats.err"
B="python bench.py --workload sim1x65536_batch_from_table --steps 3 --warmup 1"
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_bft" -o bench --output-format csv -- $B > "$OUT/bench_bft.json" 2>> "$OUT/stats.err"
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_bft_g1" -o bench --output-format csv -- $B > /dev/null 2>> "$OUT/stats.err"
timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_bft_g2" -o bench --output-format csv -- $B > /dev/null 2>> "$OUT/stats.err"
# the same chain on the sort path (DCARL_INGEST_DIRECT=0), kernel stats only: the A/B of the two ingest implementations
DCARL_INGEST_DIRECT=0 timeout -k 5 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_e2e_sort" -o bench --output-format csv -- python bench.py --workload sim1x65536_end_to_end --steps 3 --warmup 1 > "$OUT/bench_e2e_sort.json" 2>> "$OUT/stats.err"
# and on a uniformly random arrival order (both implementations), kernel stats only
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_e2e_random" -o bench --output-format csv -- python bench.py --workload sim1x65536_end_to_end --arrival-order random --steps 3 --warmup 1 > "$OUT/bench_e2e_random.json" 2>> "$OUT/stats.err"
DCARL_INGEST_DIRECT=0 timeout -k 5 400 rocprofv3 --kernel-trace --stats -d "$OUT/stats_e2e_random_sort" -o bench --output-format csv -- python bench.py --workload sim1x65536_end_to_end --arrival-order random --steps 3 --warmup 1 > "$OUT/bench_e2e_random_sort.json" 2>> "$OUT/stats.err"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/ubench_issue.hip -o tools/ubench_issue.bin > "$OUT/ubench_build.log" 2>&1   # (the binary is git-ignored: build it here)
./tools/ubench_issue.bin 4 > "$OUT/ubench_issue_4waves.txt" 2>&1 || true     # (four waves per SIMD: what the f32 online kernel runs with since round 6)
# round 5: every remaining bench leg gets its counter line (tools/pmc_legs.sh -> gpurun_out/legs_$TAG/summary)
bash tools/pmc_legs.sh "$TAG" > "$OUT/pmc_legs.log" 2>&1 || true
python tools/summarize_profile.py "$OUT" "$TAG"
# merge the legs' entries over the summary's hbm_traffic.json (leg_traffic.py was given the profiles/ copy as its base; re-base it here)
python tools/leg_traffic.py "gpurun_out/legs_$TAG" "$TAG" "$OUT/summary/hbm_traffic.json" >> "$OUT/pmc_legs.log" 2>&1 && cp gpurun_out/legs_$TAG/summary/hbm_traffic.json "$OUT/summary/hbm_traffic.json" && cp gpurun_out/legs_$TAG/summary/${TAG}_pmc_legs.csv "$OUT/summary/"
python tools/roofline_table.py "$OUT/bench.json" "$OUT/summary/hbm_traffic.json" > "$OUT/summary/${TAG}_roofline_table.md" 2>> "$OUT/bench.err" || true
# copy gpurun_out/prof_$TAG/summary/* into profiles/ (tracked) after the call returns
