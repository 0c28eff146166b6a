#!/bin/bash
# Round 5: specialised waves (form 3, the default build) against the symmetric two-accumulator form (CTG_STEM_FORM=1)
# on one slice group of the headline tree -- same library, same box; then the other m20 trees in the default form.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5_ws; mkdir -p $O
T=$R/tests/golden/trees
for form in 3 1 3 1; do
  CTG_STEM_FORM=$form timeout 300 python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --headline-only \
      --dump-steps $O/steps_f$form.json > $O/bench_f$form.out 2> $O/bench_f$form.err
  echo "form $form: $(python -c "import json; d=json.loads(open('$O/bench_f$form.out').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms/slice; dominant', d['roofline']['kernel'][-28:], round(d['roofline']['avg_launch_ms'],2), 'ms', round(d['roofline']['frac'],3))" 2>&1 | tail -1)"
done
python $R/tools/cmp_steps.py $O/steps_f1.json $O/steps_f3.json 30 > $O/cmp_f1_f3.txt 2>&1
head -70 $O/cmp_f1_f3.txt
for tree in sycamore_m20_w32_g.json sycamore_m20_w33_bf3.json; do
  for form in 3 1; do
    CTG_STEM_FORM=$form timeout 300 python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --headline-only --tree $T/$tree \
        > $O/bench_${tree%.json}_f$form.out 2> $O/bench_${tree%.json}_f$form.err
    echo "$tree form $form: $(python -c "import json; d=json.loads(open('$O/bench_${tree%.json}_f$form.out').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms/slice', round(d['est_time_total_s']), 's')" 2>&1 | tail -1)"
  done
done
