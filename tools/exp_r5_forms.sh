#!/bin/bash
# Round 5: the forms of the bf16 x 3 stem kernel side by side on one slice group of the headline tree
# (default = two-accumulator real parts + limb intermediate where it fits; xm = two accumulators only;
# r4form = round 4's form), then the SQ counters of the default build.
#   build: CTG_VARIANT_SOURCES=ctg_stem.hip python tools/build_variants.py r4form=-DCTG_STEM_FORM=0 xm=-DCTG_STEM_FORM=1
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5_forms; mkdir -p $O
TREE=${1:-sycamore_m20_native.json}
for v in default xm r4form; do
  if [ $v = default ]; then unset CTG_LIB; else export CTG_LIB=$R/cotengra_amd/lib/exp/libctg_$v.so; fi
  [ $v != default ] && [ ! -f "$CTG_LIB" ] && continue
  timeout 300 python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --headline-only \
      --tree $R/tests/golden/trees/$TREE --dump-steps $O/steps_$v.json > $O/bench_$v.out 2> $O/bench_$v.err
  echo "$v: $(python -c "import json,sys; d=json.loads(open('$O/bench_$v.out').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])")"
done
unset CTG_LIB
python $R/tools/cmp_steps.py $O/steps_r4form.json $O/steps_default.json 14 > $O/cmp_r4_default.txt 2>&1
python $R/tools/cmp_steps.py $O/steps_r4form.json $O/steps_xm.json 14 > $O/cmp_r4_xm.txt 2>&1
head -40 $O/cmp_r4_default.txt
