#!/bin/bash
# FETCH_SIZE of the final-state kernel instances on the configs[3] shape (over-fetch check)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2fetch; rm -rf $O; mkdir -p $O
for v in "4,4,2" "4,4,1" "4,6,3" "8,4,2" "16,4,2"; do
  DCARL_QUAD=$v rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O/f_$v" -o bench --output-format csv -- python bench.py --workload cfg3_sim2_argmax --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/err_$v.txt
  python - "$O/f_$v" "$v" <<'PY'
import csv, glob, sys
vals=[]
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bounds_quad" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            vals.append(float(r["Counter_Value"]))
print(sys.argv[2], "FETCH_SIZE KiB", sum(vals)/max(len(vals),1), "x2 bytes", 2*1024*sum(vals)/max(len(vals),1))
PY
done
