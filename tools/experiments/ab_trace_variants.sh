#!/bin/bash
# same-box check + A/B of tools/ab/lib<V>.so builds of the online kernels (tools/build_trace_variants.sh):
#   VARIANTS="A Q T QT" bash tools/experiments/ab_trace_variants.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
SEL='(test_trace_random_ragged_vs_oracle and (default or unsorted)) or test_trace_long or test_trace_without or test_trace_bundled or test_trace_params or test_trace_ragged_reference or test_final_table_kernel_on or test_trace_empty'
for v in ${VARIANTS:-A}; do
  [ $v = A ] && continue
  DCARL_HIP_LIB=$PWD/tools/ab/lib$v.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_resume.py -m gpu -q -x -k "$SEL or test_resume" -p no:cacheprovider 2>&1 | tail -3 | sed "s/^/$v: /"
done
for i in 1 2; do
  for v in ${VARIANTS:-A}; do
    export DCARL_HIP_LIB=$PWD/tools/ab/lib$v.so
    for w in "sim1x65536_trace" "cfg3_sim2_argmax --mode trace" "cfg4_mixed --total-states 524288 --mode trace" "sim1x65536_final_table" "dropin_a30_f64"; do
      python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$w'[:24], round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4), d['roofline'].get('kernel'))"
    done
  done
done
