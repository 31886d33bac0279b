#!/bin/bash
# SQ / LDS counters of the online kernel on configs[1], one pass per counter group, for the tree in $1 (default: this one; tools/ab/head_tree = HEAD):
#   gpurun -- 'bash tools/experiments/pmc_online.sh . new; bash tools/experiments/pmc_online.sh tools/ab/head_tree head'
cd /tmp && export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
TREE=${1:-.}; TAG=${2:-new}
OUT=$ROOT/gpurun_out/pmc_online_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT/$TREE"
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/g$i" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/g$i.err"
done
OUT_DIR=$OUT python - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, os
for f in sorted(glob.glob(os.environ["OUT_DIR"] + "/g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "trace_nwave" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
