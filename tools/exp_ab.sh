#!/bin/bash
# Same-box A/B of two libraries on the m20 trees: tools/exp_ab.sh <variant.so> [trees...]
# (alternating runs; box-to-box scatter of a slice is 2-4 %, only numbers from one box compare)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
V=$1; shift
TREES=${@:-sycamore_m20_native sycamore_m20_fused}
run() {
  env CTG_LIB=$2 timeout 300 python bench.py --headline-only --no-cpu-baseline --steps 8 --warmup 2 \
      --tree tests/golden/trees/$3.json 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $3', round(d['ms_per_step'],2), 'ms/slice', round(d['tflops'],1), 'TFLOP/s')"
}
for i in 1 2; do
  for t in $TREES; do
    run variant $R/$V $t; run current "" $t
  done
done
