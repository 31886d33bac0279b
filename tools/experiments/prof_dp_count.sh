#!/bin/bash
# per-kernel times of the direct ingest with either count pass (rocprofv3 --kernel-trace --stats of tools/experiments/prof_direct.py)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for m in queue wide; do
  DCARL_DP_COUNT=$m timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_count_$m -o p --output-format csv -- python tools/experiments/prof_direct.py 65536 1 > /dev/null 2>&1
  echo "== $m"; f=$(find gpurun_out/prof_count_$m -name 'p_kernel_stats.csv' | head -1); head -8 "$f" | cut -d, -f1-4 | cut -c1-120
done
