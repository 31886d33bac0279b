#!/bin/bash
# The direct ingest's pack kernel on an arrival order (dense | random), memory-side counters, one counter group per pass and
# every pass under its own timeout (a rocprofv3 that aborts on an uncollectable group does not exit by itself):
#   gpurun -- 'bash tools/experiments/pmc_pack_random.sh random'
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ord=${1:-random}
OUT=gpurun_out/pmc_pack_$ord
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="python bench.py --workload sim1x65536_end_to_end --arrival-order $ord --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/g$i" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/g$i.err" || echo "pass $i ($grp): rc $?"
done
python - "$OUT" <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "dp_pack" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:28s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
