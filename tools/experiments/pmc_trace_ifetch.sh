#!/bin/bash
# instruction-fetch counters of the online kernel (headline shape):  gpurun -- 'bash tools/experiments/pmc_trace_ifetch.sh'
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_ifetch; rm -rf "$OUT"; mkdir -p "$OUT"
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
i=0
for grp in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAIT_ANY" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/g$i" -o p --output-format csv -- $B > /dev/null 2> "$OUT/g$i.err"
done
python - "$OUT" trace_nwave <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:32s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
