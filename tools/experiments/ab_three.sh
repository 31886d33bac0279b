#!/bin/bash
# Same-box A/B/C of three builds on the headline: A = tools/ab/libA.so, B = in-tree, C = tools/ab/libC.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2 3; do
  for v in A B C; do
    case $v in A) export DCARL_HIP_LIB=$PWD/tools/ab/libA.so;; B) unset DCARL_HIP_LIB;; C) export DCARL_HIP_LIB=$PWD/tools/ab/libC.so;; esac
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['roofline']['kernel'], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
  done
done
