#!/bin/bash
# where the final-state kernel's 1.19x over-fetch on the configs[3] shape comes from.  Separate counter passes
# (TCC slots are few) over  bench.py --workload cfg3_sim2_argmax  (2^20 states, 364-byte unaligned buckets):
#   requests the L1s (TCP) send to the L2 (TCC), L2 hits / misses, and the L2's read requests to memory by size.
#   gpurun -- 'tools/experiments/pmc_cfg3_overfetch.sh r04'   ->  gpurun_out/overfetch_<tag>/summary.csv
set -u
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/overfetch_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
B="python bench.py --workload cfg3_sim2_argmax --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs ${2:-}"
i=0
for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum" \
           "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_READ_SECTORS_sum TCC_STREAMING_REQ_sum" \
           "SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d "$OUT/g$i" -o p --output-format csv -- $B > "$OUT/bench_g$i.json" 2>> "$OUT/err.txt"
done
python - "$OUT" <<'PY'
import sys, glob, csv, json, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "bounds_quad" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
t = {k: sum(v) / len(v) for k, v in acc.items()}
line = json.loads([l for l in open(out + "/bench_g1.json") if l.startswith("{")][-1])
alg = line["roofline"]["algorithmic_bytes"]
rows = [("algorithmic_bytes", alg)] + sorted(t.items())
d = dict(rows)
if "FETCH_SIZE" in d: rows.append(("fetch_bytes = 2*FETCH_SIZE*1024", 2 * d["FETCH_SIZE"] * 1024))
if "TCC_EA0_RDREQ_sum" in d:
    r32, r64, r128 = d.get("TCC_EA0_RDREQ_32B_sum", 0), d.get("TCC_EA0_RDREQ_64B_sum", 0), d.get("TCC_EA0_RDREQ_128B_sum", 0)
    rows.append(("ea_read_bytes = 32*RD32 + 64*RD64 + 128*RD128", 32 * r32 + 64 * r64 + 128 * r128))
with open(out + "/summary.csv", "w") as f:
    f.write("quantity,mean per launch of bounds_quad_kernel\n")
    for k, v in rows: f.write(f"{k},{v:.0f}\n")
print(open(out + "/summary.csv").read())
PY
