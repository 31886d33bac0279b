for cfg in "single 0" "single 20000" "single 33000" "pair 0" "pair 14000" "pair 27000" "single 0" "pair 14000"; do
  set -- $cfg
  DCARL_TRACE_KERNEL=$1 DCARL_LDS_PAD=$2 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2', round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))"
done
