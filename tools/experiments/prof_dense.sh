# per-kernel times of the direct ingest on the configs[1] table (dense order): gpurun -- 'bash tools/experiments/prof_dense.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dense -o p --output-format csv -- python tools/experiments/prof_direct.py 65536 1 > $R/gpurun_out/prof_dense.log 2>&1)
python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/prof_dense/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    if "dp_" in r['Name']: print(f"{r['Name'].split('::')[-1][:40]:40s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
