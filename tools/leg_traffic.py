#!/usr/bin/env python3
"""Per-step HBM bytes of the bench legs profiled by tools/pmc_legs.sh: for every leg the FETCH_SIZE and WRITE_SIZE passes over the
same command, bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per dispatch (MI355X_MICROARCH.md, HBM section: KiB units, FETCH_SIZE
counts half of a wide stream on gfx950 — cross-checked against TCC_EA0_RDREQ_* in round 4), summed over the leg's kernels and
divided by the number of steps (chains) the run executed.  Writes <tag>_pmc_legs.csv (leg, kernel, launches per step, bytes) and
merges "<traffic key>|<algorithmic bytes>" entries into hbm_traffic.json, which is what bench.py's load_traffic() reads.

    python tools/leg_traffic.py gpurun_out/legs_r05 r05 [profiles/hbm_traffic.json]"""
import collections
import csv
import glob
import json
import os
import re
import sys

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(src, "summary")
os.makedirs(dst, exist_ok=True)
base_json = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "hbm_traffic.json")

CHAIN = r"dp_|slots_kernel|slice_|lengths_|unit_slice|trace_nwave|pad_"
# leg -> (traffic key of bench.py, kernels of a step, the kernel whose dispatch count is the number of steps run; None: the JSON's)
LEGS = {
    "e2e_random": ("end_to_end_random", CHAIN, "dp_partition"),
    "e2e_dense": ("end_to_end", CHAIN, "dp_partition"),
    "bft": ("batch_from_table", CHAIN + r"|final_table_kernel|bounds_", "dp_partition"),
    "buckets": ("buckets_from_table", CHAIN + r"|count_records|regroup|bounds_", "dp_partition"),
    "final_table": ("final_table_kernel", r"final_table_kernel", "final_table_kernel"),
    "pairs_1e6": ("sample_pairs_kernel", r"sample_pairs_kernel", "sample_pairs_kernel"),
    "pairs_2p30": ("sample_pairs_kernel", r"sample_pairs_kernel", "sample_pairs_kernel"),
    "s2e": ("sampler_to_estimator", r"sample_pairs_kernel|" + CHAIN, "sample_pairs_kernel"),
    "s2l": ("sampler_into_layout", r"sample_state_records_ragged|rx_|slots_kernel|slice_|lengths_|trace_nwave", "sample_state_records_ragged"),
    "host_streamed": ("host_streamed", r"dcarl", None),
}


def short(k):
    k = re.sub(r"^void\s+", "", k)
    k = k.replace("dcarl::(anonymous namespace)::", "").replace("dcarl::", "")
    return re.sub(r"\(.*$", "", k)[:90]


def counters(d, name):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                a = acc[r["Kernel_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return acc


rows, traffic = [], {}
for leg, (key, pat, anchor) in LEGS.items():
    jf = os.path.join(src, leg + ".json")
    try:
        line = [l for l in open(jf).read().splitlines() if l.startswith("{")][-1]
        bj = json.loads(line)
    except Exception as e:   # noqa: BLE001
        print(f"{leg}: no bench line ({e!r})")
        continue
    fe, wr = counters(os.path.join(src, leg + "_fetch"), "FETCH_SIZE"), counters(os.path.join(src, leg + "_write"), "WRITE_SIZE")
    if not fe or not wr:
        print(f"{leg}: a counter pass is missing")
        continue
    # steps PER PASS: legs with a time-based settle phase launch a different number of steps in the two passes
    if anchor is None:
        steps_f = steps_w = bj["steps"]
    else:
        steps_f = max(v[0] for k, v in fe.items() if anchor in k)
        steps_w = max(v[0] for k, v in wr.items() if anchor in k)
    steps = steps_f
    total = 0.0
    for k in sorted(set(fe) | set(wr)):
        if "dcarl" not in k or not re.search(pat, k):
            continue
        f_b, w_b = 2 * fe.get(k, [0, 0.0])[1] * 1024.0 / steps_f, wr.get(k, [0, 0.0])[1] * 1024.0 / steps_w
        calls = fe.get(k, [0])[0] / steps_f
        total += f_b + w_b
        rows.append(dict(leg=leg, kernel=short(k), launches_per_step=round(calls, 3), fetch_bytes=f_b, write_bytes=w_b, hbm_bytes=f_b + w_b))
    alg = bj["roofline"]["algorithmic_bytes"]
    rows.append(dict(leg=leg, kernel="== step total ==", launches_per_step="", fetch_bytes="", write_bytes="", hbm_bytes=total))
    rows.append(dict(leg=leg, kernel="== algorithmic ==", launches_per_step="", fetch_bytes="", write_bytes="", hbm_bytes=alg))
    traffic[f"{key}|{int(alg)}"] = dict(algorithmic_bytes=int(alg), hbm_bytes_per_launch=total, traffic_over_algorithmic=total / alg,
                                        steps_in_profiled_run=steps,
                                        correction="sum over the step's kernels of (2*FETCH_SIZE + WRITE_SIZE) * 1024, separate passes",
                                        workload=bj["config"]["workload"], source=f"profiles/{tag}_pmc_legs.csv ({leg})")
    print(f"{leg:14s} steps {steps:3d}  hbm {total / 1e9:9.3f} GB  algorithmic {alg / 1e9:9.3f} GB  x{total / alg:.3f}  {bj['ms_per_step']:.3f} ms")
with open(os.path.join(dst, f"{tag}_pmc_legs.csv"), "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=["leg", "kernel", "launches_per_step", "fetch_bytes", "write_bytes", "hbm_bytes"])
    w.writeheader()
    w.writerows(rows)
try:
    merged = json.load(open(base_json))
except Exception:   # noqa: BLE001
    merged = {}
merged.update(traffic)
json.dump(merged, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
