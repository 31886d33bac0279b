#!/usr/bin/env python3
"""Turn a tools/profile_round.sh output directory into the small summaries committed under profiles/."""
import glob
import json
import os
import sys

import pandas as pd

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(src, "summary")
os.makedirs(dst, exist_ok=True)


def find(sub, pat):
    hits = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


out = {}
st = find("stats", "*kernel_stats.csv")
if st:
    d = pd.read_csv(st)
    d.to_csv(os.path.join(dst, f"{tag}_kernel_stats.csv"), index=False)
    out["kernel_stats_top"] = d.head(6).to_dict(orient="records")
kt = find("stats", "*kernel_trace.csv")
if kt:
    k = pd.read_csv(kt)
    k["dur_us"] = (k.End_Timestamp - k.Start_Timestamp) / 1e3
    g = k.groupby("Kernel_Name").dur_us.agg(["count", "mean", "min", "max", "sum"]).sort_values("sum", ascending=False)
    g.to_csv(os.path.join(dst, f"{tag}_kernel_trace_summary.csv"))
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    if f:
        c = pd.read_csv(f)
        c = c[c.Counter_Name == ctr]
        g = c.groupby("Kernel_Name").Counter_Value.agg(["count", "mean"]).sort_values("mean", ascending=False)
        g.to_csv(os.path.join(dst, f"{tag}_pmc_{ctr}.csv"))
        out[ctr] = {k[:60]: v for k, v in g["mean"].head(4).to_dict().items()}
f = find("sq", "*counter_collection.csv")
if f:
    c = pd.read_csv(f)
    g = c.groupby(["Kernel_Name", "Counter_Name"]).Counter_Value.mean().unstack()
    g.to_csv(os.path.join(dst, f"{tag}_pmc_SQ.csv"))
ow = os.path.join(src, "other_workloads.jsonl")
if os.path.exists(ow):
    lines = [l for l in open(ow).read().splitlines() if l.startswith("{")]
    open(os.path.join(dst, f"{tag}_other_workloads.jsonl"), "w").write("\n".join(lines) + "\n")
    out["other_workloads"] = [dict(workload=json.loads(l)["config"]["workload"], value=json.loads(l)["value"],
                                   unit=json.loads(l)["unit"], kernel_ms=json.loads(l)["roofline"]["kernel_ms"],
                                   frac=json.loads(l)["roofline"]["frac"]) for l in lines]
sb = find("stats_batch", "*kernel_stats.csv")
if sb:
    pd.read_csv(sb).head(8).to_csv(os.path.join(dst, f"{tag}_kernel_stats_batch.csv"), index=False)
for name in ("bench.json", "bench_stats.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        try:
            out[name] = json.loads(open(p).read().strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            out[name] = f"unparsed: {e}"
# HBM traffic per launch of the dominant kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
# prescribes: the counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced stream
# (x2); WRITE_SIZE is calibrated here on sample_state_records_kernel, whose written byte count is known exactly.
try:
    bench = out.get("bench.json") or out.get("bench_stats.json")
    kern = bench["roofline"]["kernel"].split("<")[0]
    f = pd.read_csv(os.path.join(dst, f"{tag}_pmc_FETCH_SIZE.csv")).set_index("Kernel_Name")["mean"]
    w = pd.read_csv(os.path.join(dst, f"{tag}_pmc_WRITE_SIZE.csv")).set_index("Kernel_Name")["mean"]
    fk = [k for k in f.index if kern in k][0]
    wk = [k for k in w.index if kern in k][0]
    cal = [k for k in w.index if "sample_state_records" in k]
    cfg = bench["config"]
    known = cfg["states_per_gpu"] * ((cfg["records_per_state"] + 3) // 4 * 4) * 5 if cal else None
    traffic = {kern: dict(algorithmic_bytes=bench["roofline"]["algorithmic_bytes"],
                          fetch_size_kib=float(f[fk]), write_size_kib=float(w[wk]),
                          hbm_bytes_per_launch=float((2.0 * f[fk] + w[wk]) * 1024.0),
                          correction="bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
                          write_calibration=(dict(kernel="sample_state_records_kernel", known_bytes=known,
                                                  counter_bytes=float(w[cal[0]] * 1024.0)) if cal else None),
                          source=f"profiles/{tag}_pmc_FETCH_SIZE.csv, profiles/{tag}_pmc_WRITE_SIZE.csv")}
    json.dump(traffic, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
    out["hbm_traffic"] = traffic
except Exception as e:  # noqa: BLE001
    out["hbm_traffic"] = f"not derived: {e!r}"
json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
