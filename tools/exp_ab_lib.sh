#!/bin/bash
# same-box alternating A/B of two builds of the library on the headline tree: tools/exp_ab_lib.sh <variant.so> [tree]
R=${GRAFT_REPO_ROOT:-$PWD}; V=$1; TREE=${2:-sycamore_m20_native.json}
for v in default variant default variant; do
  if [ $v = default ]; then unset CTG_LIB; else export CTG_LIB=$R/$V; fi
  timeout 900 python $R/bench.py --steps 8 --warmup 2 --headline-only --no-cpu-baseline --tree $R/tests/golden/trees/$TREE 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'ms/slice', d['ms_per_step'], 'TF', round(d['value']/1e12,1), 'dominant', d['roofline']['avg_launch_ms'])"
done
