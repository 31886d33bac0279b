#!/bin/bash
# SQ counters of one radix pass (tools/experiments/ubench_scatter_lines.bin): VALU / LDS busy, bank conflicts
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_lines; rm -rf $OUT; mkdir -p $OUT
B="tools/experiments/ubench_scatter_lines.bin 268435456 5 8"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o p --output-format csv -- $B > /dev/null 2> $OUT/g$i.err
done
python - <<'PY'
import csv, glob, collections
for g in sorted(glob.glob('gpurun_out/pmc_lines/g*/p_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(g)):
        k = r['Kernel_Name']
        if 'rx_scatter' not in k: continue
        name = 'lines<' + k.split('rx_scatter_lines_kernel<')[1].split('>')[0] + '>' if 'lines' in k else 'old'
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
    for name, cs in acc.items():
        print(name, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
