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
for name in ("bench.json", "bench_stats.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        try:
            out[name] = json.loads(open(p).read().strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            out[name] = f"unparsed: {e}"
json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
