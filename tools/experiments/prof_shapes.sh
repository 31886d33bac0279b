# per-kernel times of one direct-path ingest on table shapes (prof_direct_shape.py): gpurun -- 'bash tools/experiments/prof_shapes.sh ["N S kind" ...]'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ $# -eq 0 ]; then set -- "1310720000 65536 skewed" "1310720000 65536 uniform"; fi
for spec in "$@"; do
  set -- $spec
  tag=$3_$1
  (cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_shape_$tag -o p --output-format csv -- python tools/experiments/prof_direct_shape.py $1 $2 $3 1 > $R/gpurun_out/prof_shape_$tag.log 2>&1)
  echo "== $tag"; python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/prof_shape_$tag/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    if "dcarl" in r['Name']: print(f"{r['Name'].split('(')[1 if r['Name'].startswith('dcarl::(') else 0][:50]:50s} {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
done
