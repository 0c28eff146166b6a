#!/bin/bash
# same-box A/B of two builds of the library on the headline tree, per step: tools/exp_ab_steps.sh <variant.so> [tree]
R=${GRAFT_REPO_ROOT:-$PWD}; V=$1; TREE=${2:-sycamore_m20_native.json}; O=$R/gpurun_out/ab_steps; mkdir -p $O
for v in default variant default variant; do
  if [ $v = default ]; then unset CTG_LIB; else export CTG_LIB=$R/$V; fi
  timeout 900 python $R/bench.py --steps 4 --warmup 1 --headline-only --no-cpu-baseline --tree $R/tests/golden/trees/$TREE \
      --dump-steps $O/steps_$v.json 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'ms/slice', d['ms_per_step'], 'TF', round(d['value']/1e12,1), 'dominant', d['roofline']['avg_launch_ms'])"
done
unset CTG_LIB
python $R/tools/cmp_steps.py $O/steps_default.json $O/steps_variant.json 24 2>&1 | grep "^  #" | cut -c1-250
