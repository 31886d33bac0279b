#!/bin/bash
# same-box A/B of the working tree against the committed HEAD (tools/ab/head_tree: `git archive HEAD` + its built library):
#   bash tools/experiments/ab_vs_head.sh ["workload args" ...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
W=("$@")
[ ${#W[@]} -eq 0 ] && W=("sim1x65536_trace" "cfg3_sim2_argmax --mode trace" "cfg4_mixed --total-states 524288 --mode trace" "dropin_a30_f64")
for i in 1 2; do
  for v in HEAD NEW; do
    for w in "${W[@]}"; do
      if [ $v = HEAD ]; then d=tools/ab/head_tree; else d=.; fi
      (cd $d && python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null) | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$w'[:24], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4), d['roofline'].get('kernel'))"
    done
  done
done
