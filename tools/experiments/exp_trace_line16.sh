#!/bin/bash
# Round 5: the online kernel with its INPUT arrays (L1; L2: the per-record outputs too) addressed in a layout whose
# unit per state is a whole 64-byte line (16 records) instead of a 16-byte quad — timing only (the tables keep the quad layout, so
# the values are not the table's; all states are statistically alike).  Same-box A/B of separately built libraries:
#   tools/build_variant.sh tools/ab/libA.so; ... libL1.so -DDCARL_TRACE_LINE16=1; libL2.so -DDCARL_TRACE_LINE16=2;
#   libL1n.so -DDCARL_TRACE_LINE16=1 -DDCARL_TRACE_NT=1 (temporal loads)
#   gpurun -- 'bash tools/experiments/exp_trace_line16.sh'   -> gpurun_out/line16/
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/line16; rm -rf $OUT; mkdir -p $OUT
for i in 1 2 3; do
  for v in A L1 L1n L2; do
    export DCARL_HIP_LIB=$PWD/tools/ab/lib$v.so
    for w in "sim1x65536_trace" "cfg3_sim2_argmax --mode trace" "cfg4_mixed --total-states 524288 --mode trace"; do
      python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$w'[:24], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
    done
  done
done | tee $OUT/timing.txt
for v in A L1 L1n L2; do
  export DCARL_HIP_LIB=$PWD/tools/ab/lib$v.so
  B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
  for c in FETCH_SIZE WRITE_SIZE "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    timeout -k 5 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_${v}_$n -o p --output-format csv -- $B > /dev/null 2> $OUT/pmc_${v}_$n.err
  done
done
python - <<'PY'
import csv, glob, os, collections
out = "gpurun_out/line16"
rows = []
for d in sorted(glob.glob(out + "/pmc_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("void dcarl::trace_nwave_kernel"):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            rows.append((os.path.basename(d), k, len(v), sum(v) / len(v)))
with open(out + "/counters.txt", "w") as fh:
    for r in rows:
        print(*r, file=fh); print(*r)
PY
