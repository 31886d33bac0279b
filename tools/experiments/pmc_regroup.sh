#!/bin/bash
# counters of regroup_kernel (group_records, the write-combining form) on the configs[1] table:  gpurun -- 'bash tools/experiments/pmc_regroup.sh'
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_regroup; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/rg.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import dcarl_amd as dc
S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
t = dc.sampler.sample_state_records(dc.workloads.sim1_q_row(), 20000, seed=0, stream_id=0, S=S)
for _ in range(3):
    v, seg = t.to_buckets()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
lib, P = dc._lib.load(), dc._lib.ptr
e0.record()
for _ in range(3):
    dc._lib.check(lib.dcarl_group_records_f32(P(t.R), P(t.act), P(t.slice_row_off), P(t.lengths), P(t.slot_state_i32), t.S, t.A, P(seg), P(v), dc._lib.stream_ptr()))
e1.record(); torch.cuda.synchronize()
print("S", S, "regroup ms", e0.elapsed_time(e1) / 3)
PY
for S in 65536 49152 32768 16384; do python /tmp/rg.py $S 2>/dev/null; done | tee $OUT/timing.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU" \
  "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o p --output-format csv -- python /tmp/rg.py 65536 > /dev/null 2> $OUT/g$i.err || echo "pass $i: rc $?"
done
python - $OUT <<'PY' | tee -a $OUT/timing.txt
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "regroup" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:28s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
