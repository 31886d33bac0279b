#!/bin/bash
# same-box A/B of the final-state kernel instances: DCARL_QUAD=G,NV,U (U >= 1: plain, U buckets per cluster per pass;
# U = 0 / -D: software-pipelined with D register buffers).  Two interleaved rounds; compare the per-variant minimum.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r2g}; mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"; }
timeout 900 python -m pytest tests/test_configs_full.py tests/test_episodes.py -x -q -m gpu > $O/t_new.log 2>&1; echo "new tests rc=$?" >> $O/t_new.log
for round in 1 2; do
for v in ${VARIANTS:-"4,4,2" "8,4,2" "4,4,1" "4,6,3"}; do
  for w in "cfg3_sim2_argmax" "cfg4_mixed --total-states 524288" "sim1x65536_batch"; do
    echo -n "DCARL_QUAD=$v $w : " >> $O/ab.log
    DCARL_QUAD=$v timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>>$O/ab.err | line >> $O/ab.log 2>&1
  done
done
done
tail -3 $O/t_new.log; sort $O/ab.log
