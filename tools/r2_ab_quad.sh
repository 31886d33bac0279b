#!/bin/bash
# same-box A/B of the bounds_quad_kernel instances (G lanes per bucket, NV vector slots) + SQ counters of the default one
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2b; mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"; }
timeout 900 python -m pytest tests/test_configs_full.py -x -q -m gpu > $O/t_new.log 2>&1; echo "new tests rc=$?" >> $O/t_new.log
for v in "4,8" "4,4" "4,6" "4,12" "8,4" "8,8" "16,4" "16,8"; do
  for w in "cfg3_sim2_argmax" "cfg4_mixed --total-states 524288" "sim1x65536_batch"; do
    echo -n "DCARL_QUAD=$v $w : " >> $O/ab.log
    DCARL_QUAD=$v DCARL_BOUNDS_KERNEL=quad timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>>$O/ab.err | line >> $O/ab.log 2>&1
  done
done
for k in rows csr64; do
  echo -n "$k sim1x65536_batch : " >> $O/ab.log
  DCARL_BOUNDS_KERNEL=$k timeout 600 python bench.py --workload sim1x65536_batch --steps 10 --warmup 2 --no-cpu-baseline 2>>$O/ab.err | line >> $O/ab.log 2>&1
done
BENCH="python bench.py --workload cfg3_sim2_argmax --steps 3 --warmup 1 --no-cpu-baseline"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d "$O/g$i" -o bench --output-format csv -- $BENCH > /dev/null 2> "$O/g$i.err"
done
python - "$O" "bounds_quad" <<'PY' > $O/pmc_quad_cfg3.txt
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
tail -3 $O/t_new.log; cat $O/ab.log; cat $O/pmc_quad_cfg3.txt
