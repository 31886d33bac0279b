#!/bin/bash
# same-box A/B of the bounds_quad_kernel instances (G lanes per bucket, NV vector slots, U buckets per cluster per pass)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2c; mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"; }
timeout 900 python -m pytest tests/test_configs_full.py tests/test_episodes.py -x -q -m gpu > $O/t_new.log 2>&1; echo "new tests rc=$?" >> $O/t_new.log
for v in "4,8,1" "4,6,1" "4,4,2" "4,6,2" "4,8,2" "8,4,2" "4,4,3" "4,6,3" "4,4,4" "4,8,1"; do
  for w in "cfg3_sim2_argmax" "cfg4_mixed --total-states 524288" "sim1x65536_batch"; do
    echo -n "DCARL_QUAD=$v $w : " >> $O/ab.log
    DCARL_QUAD=$v DCARL_BOUNDS_KERNEL=quad timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>>$O/ab.err | line >> $O/ab.log 2>&1
  done
done
tail -3 $O/t_new.log; cat $O/ab.log
