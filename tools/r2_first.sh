#!/bin/bash
# round 2, first GPU call: new tests, whole GPU suite, A/B of the final-state kernels on configs[3]/[4], default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r2a
O=gpurun_out/r2a
timeout 1500 python -m pytest tests/test_configs_full.py -x -q -m gpu > $O/t_new.log 2>&1; echo "new tests rc=$?" >> $O/t_new.log
for k in quad rows; do
  for w in "cfg3_sim2_argmax" "cfg4_mixed --total-states 524288" "sim1x65536_batch"; do
    echo "== $k $w" >> $O/ab.log
    DCARL_BOUNDS_KERNEL=$k timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>>$O/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4), d['value'])" >> $O/ab.log 2>&1
  done
done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/bench_default.err
timeout 1800 python -m pytest tests -x -q -m gpu --deselect tests/test_configs_full.py > $O/t_all.log 2>&1; echo "all tests rc=$?" >> $O/t_all.log
tail -5 $O/t_new.log; cat $O/ab.log; tail -3 $O/t_all.log; tail -c 1500 $O/bench_default.json
