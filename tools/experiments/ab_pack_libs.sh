#!/bin/bash
# per-kernel times of the direct ingest, library A (tools/ab/libA.so via DCARL_HIP_LIB) against the in-tree build, dense and random order
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
for v in A B; do
  if [ $v = A ]; then export DCARL_HIP_LIB=$PWD/tools/ab/libA.so; else unset DCARL_HIP_LIB; fi
  timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_pack_$v -o p --output-format csv -- python tools/experiments/prof_direct.py 65536 1 > /dev/null 2>&1
  f=$(find gpurun_out/prof_pack_$v -name 'p_kernel_stats.csv' | head -1)
  echo "== $v (rep $rep): $(grep -E 'dp_pack_kernel|dp_partition' $f | awk -F, '{printf "%s %.3f ms  ", substr($1,1,60), $4/1e6}')"
done
done
unset DCARL_HIP_LIB
python tools/experiments/bench_e2e_quick.py 2>&1 | grep -v amdgpu.ids | grep direct
