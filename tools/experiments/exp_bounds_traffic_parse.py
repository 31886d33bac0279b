"""counter csv of exp_bounds_traffic.py -> read traffic over sample bytes per table"""
import glob, json, sys
import pandas as pd
plan = json.load(open("gpurun_out/exp_bounds_traffic_plan.json"))
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
df = pd.read_csv(f)
df = df[df["Kernel_Name"].str.contains("bounds_") & (df["Counter_Name"] == sys.argv[2])].sort_values("Dispatch_Id")
vals = df["Counter_Value"].to_numpy()
names = df["Kernel_Name"].to_numpy()
i = 0
for p in plan:
    v = vals[i:i + p["launches"]]
    k = names[i][:60]
    i += p["launches"]
    b = 2 * v.mean() * 1024 if sys.argv[2] == "FETCH_SIZE" else v.mean() * 1024
    print(f"{p['name']:14s} {p['ms']:.3f} ms  {sys.argv[2]} bytes {b / 1e9:.3f} GB  samples {p['sample_bytes'] / 1e9:.3f} GB  offsets {p['seg_bytes'] / 1e9:.3f} GB  "
          f"ratio to samples+offsets {b / (p['sample_bytes'] + p['seg_bytes']):.3f}  {k}")
