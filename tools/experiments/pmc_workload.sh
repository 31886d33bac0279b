#!/bin/bash
# SQ counters of any bench workload:  gpurun -- 'bash tools/experiments/pmc_workload.sh cfg3_sim2_argmax bounds_quad'
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
W=$1; PAT=$2
OUT=gpurun_out/pmc_$W
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d "$OUT/g$i" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/g$i.err"
done
python - "$OUT" "$PAT" <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
