#!/bin/bash
# whole GPU suite + the default bench line + same-box repeats of the headline (box-to-box variance is +-5 %)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r2h}; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; echo "all tests rc=$?" >> $O/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/bench_default.err
for k in trio duo trio; do
  DCARL_TRACE_KERNEL=$k python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$k', d['roofline']['kernel'], round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))" >> $O/ab_trace.log
done
tail -4 $O/t_all.log; cat $O/ab_trace.log
python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('headline', d['roofline']['kernel'], round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4), d['value'])
for k,v in d['other_configs'].items():
    print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in('kernel','kernel_ms','frac','error')})
PY
