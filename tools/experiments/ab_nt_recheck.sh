#!/bin/bash
# re-check on another box: online kernel without non-temporal accesses (T0), partition without non-temporal stores (D0), sampler without (S0), in-tree (B)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -c "
import torch,sys
sys.path.insert(0,'.')
import bench
print('copy GB/s', round(bench.measured_copy_gbs()))" 2>/dev/null
for i in 1 2 3; do
  for v in T0 B; do
    if [ $v = B ]; then unset DCARL_HIP_LIB; else export DCARL_HIP_LIB=$PWD/tools/ab/lib$v.so; fi
    for w in "sim1x65536_trace" "cfg3_sim2_argmax --mode trace"; do
      python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$w'[:40].ljust(40), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
    done
  done
  for v in D0 B; do
    if [ $v = B ]; then unset DCARL_HIP_LIB; else export DCARL_HIP_LIB=$PWD/tools/ab/lib$v.so; fi
    for w in "sim1x65536_end_to_end" "sim1x65536_end_to_end --arrival-order random"; do
      python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$w'[:44].ljust(44), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
    done
  done
done
