#!/bin/bash
# Round 6: what halving the product count of the bf16 x 3 stem kernel is worth in TIME (the experiment build
# -DCTG_STEM_KO_HALF keeps the three products a two-limb split would keep and drops the third limbs: its results
# are those of a 16-bit split, the timing is that of VERDICT r5's lever (b), two fp16 limbs and three products).
#   build: CTG_VARIANT_SOURCES=ctg_stem.hip python tools/build_variants.py half=-DCTG_STEM_KO_HALF
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6_half; mkdir -p $O
TREE=${1:-sycamore_m20_native.json}
for v in default half; do
  if [ $v = default ]; then unset CTG_LIB; else export CTG_LIB=$R/cotengra_amd/lib/exp/libctg_$v.so; fi
  [ $v != default ] && [ ! -f "$CTG_LIB" ] && continue
  timeout 600 python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --headline-only \
      --tree $R/tests/golden/trees/$TREE > $O/bench_$v.out 2> $O/bench_$v.err
  echo "$v: $(python -c "import json,sys; d=json.loads(open('$O/bench_$v.out').read().strip().splitlines()[-1]); print('ms/slice', d['ms_per_step'], 'TF', d['value']/1e12, 'dominant', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])")"
  (rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | head -4) > $O/smi_$v.txt
done
unset CTG_LIB
