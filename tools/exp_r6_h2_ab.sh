#!/bin/bash
# same-box A/B of the fp16 x 2 rules on the headline tree: default (first pair of a stem in bf16 x 3) vs CTG_STEM_H2_ALL=1
# (a max-abs pass supplies the first pair's scale) vs bf16 x 3 everywhere
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6_h2ab; mkdir -p $O
TREE=${1:-sycamore_m20_native.json}
for v in default all bf16x3 default all; do
  unset CTG_STEM_H2_ALL CTG_STEM_ARITH
  [ $v = all ] && export CTG_STEM_H2_ALL=1
  [ $v = bf16x3 ] && export CTG_STEM_ARITH=bf16x3
  timeout 900 python $R/bench.py --steps 8 --warmup 2 --headline-only --no-cpu-baseline \
      --tree $R/tests/golden/trees/$TREE > $O/bench_$v.out 2> $O/bench_$v.err
  echo "$v: $(python -c "import json,sys; d=json.loads(open('$O/bench_$v.out').read().strip().splitlines()[-1]); print('ms/slice', d['ms_per_step'], 'TF', round(d['value']/1e12,1), 'dominant', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])")"
done
