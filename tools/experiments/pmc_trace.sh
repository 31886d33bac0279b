#!/bin/bash
# SQ / LDS counters of the trace kernel (one pass per counter group):  gpurun -- 'bash tools/experiments/pmc_trace.sh tab'
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export DCARL_TRACE_KERNEL=${1:-trio}
OUT=gpurun_out/pmc_$DCARL_TRACE_KERNEL
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d "$OUT/g$i" -o bench --output-format csv -- $BENCH > /dev/null 2> "$OUT/g$i.err"
done
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT_DIR")
for f in sorted(glob.glob("gpurun_out/pmc_%s/g*/**/*counter_collection.csv" % os.environ["DCARL_TRACE_KERNEL"], recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "trace" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
