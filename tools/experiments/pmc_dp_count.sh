#!/bin/bash
# SQ / LDS counters of the direct ingest's count pass in either form:  gpurun -- 'bash tools/experiments/pmc_dp_count.sh'
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for m in queue wide; do
  OUT=gpurun_out/pmc_count_$m
  rm -rf "$OUT"; mkdir -p "$OUT"
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    DCARL_DP_COUNT=$m timeout -k 5 300 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/g$i" -o p --output-format csv -- python tools/experiments/prof_direct.py 65536 1 > /dev/null 2> "$OUT/g$i.err"
  done
  echo "== $m"
  python - "$OUT" dp_count <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
done
