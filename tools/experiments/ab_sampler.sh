#!/bin/bash
# Same-box A/B of two library builds on configs[2] at both sizes (A = tools/ab/libA.so, B = in-tree); prints kernel ms, frac, in-graph ms
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then export DCARL_HIP_LIB=$PWD/tools/ab/libA.so; else unset DCARL_HIP_LIB; fi
    for r in 1000000 1073741824; do
      python bench.py --workload sampler_pairs --records $r --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); g=d['roofline'].get('in_hip_graph') or d.get('in_hip_graph') or {}; print('$v', $r, round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4), 'graph', g.get('kernel_ms'))"
    done
  done
done
