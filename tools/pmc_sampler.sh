#!/bin/bash
# memory-side counters of sample_pairs_kernel at 2^30 pairs, one group per pass:
#   gpurun -- 'bash tools/pmc_sampler.sh <k>'  -> gpurun_out/pmc_sampler_<k>.txt
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
K=${1:-0}
OUT=gpurun_out/pmc_sampler_$K; rm -rf $OUT; mkdir -p $OUT
B="python bench.py --workload sampler_pairs --records 1073741824 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs"
i=0
for grp in "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR" "TCC_EA0_WR_UNCACHED_32B_sum TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum"; do
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o p --output-format csv -- $B > $OUT/g$i.json 2> $OUT/g$i.err || echo "pass $i ($grp): rc $?"
done
python - $OUT <<'PY' | tee gpurun_out/pmc_sampler_$K.txt
import csv, glob, collections, sys, json
for f in sorted(glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "sample_pairs" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:36s} {sum(v)/len(v):18.0f}  (n={len(v)})")
for f in sorted(glob.glob(sys.argv[1] + "/g*.json")):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
        print(f, "kernel_ms", round(d["roofline"]["kernel_ms"], 4))
    except Exception as e:
        print(f, "no line", repr(e))
PY
