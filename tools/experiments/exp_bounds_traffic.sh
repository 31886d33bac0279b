#!/bin/bash
# FETCH_SIZE of the final-state kernel on uniform / random CSR tables, one rocprofv3 pass per library:
#   LIBS="tools/ab/head_tree/dcarl_amd/libdcarl_hip.so tools/ab/libB0.so tools/ab/libB1.so" bash tools/experiments/exp_bounds_traffic.sh
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for L in $LIBS; do
  echo "== $L"
  rm -rf gpurun_out/exp_bt
  DCARL_HIP_LIB=$PWD/$L timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/exp_bt -o x --output-format csv -- python tools/experiments/exp_bounds_traffic.py > /dev/null 2>&1
  python tools/experiments/exp_bounds_traffic_parse.py gpurun_out/exp_bt FETCH_SIZE | cut -c1-150
done
