#!/bin/bash
# Round 6: the fp16 x 2 arithmetic of the stem kernels against bf16 x 3 and fp32 on the headline tree (same box):
# ms per slice, executed TFLOP/s, the dominant pair's launch time and roofline fraction, and the precision checks
# of bench.py (one narrowed slice; a 4096-slice device sum against one wide complex128 oracle slice).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6_h2; mkdir -p $O
TREE=${1:-sycamore_m20_native.json}
for v in fp16x2 bf16x3 fp32; do
  CTG_STEM_ARITH=$v timeout 900 python $R/bench.py --steps 8 --warmup 2 --headline-only \
      --tree $R/tests/golden/trees/$TREE > $O/bench_$v.out 2> $O/bench_$v.err
  echo "$v: $(python -c "import json,sys; d=json.loads(open('$O/bench_$v.out').read().strip().splitlines()[-1]); print('ms/slice', d['ms_per_step'], 'TF', round(d['value']/1e12,1), 'dominant', d['roofline']['kernel'][:14], d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'precision', d.get('precision'))")"
done
