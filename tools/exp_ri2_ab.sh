#!/bin/bash
# A/B of the fused stem kernel's step 2 on ONE box: the round-3 kernel (X / Y form everywhere;
# cotengra_amd/lib/exp/libctg_r3stem.so = this round's library with csrc/ctg_stem.hip of commit
# 31948e9) against this round's (row-interleaved form where it applies), alternating, native and
# fused tree, fp32 products and bf16 x 3.  Box-to-box scatter of a slice is 2-4 %: only numbers
# from one box compare.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
run() {  # lib-tag tree env
  local lib=""; [ "$1" = r3 ] && lib=$R/cotengra_amd/lib/exp/libctg_r3stem.so
  env CTG_LIB=$lib $3 timeout 300 python bench.py --headline-only --no-cpu-baseline --steps 8 --warmup 2 \
      --tree tests/golden/trees/$2.json 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2 $3', round(d['ms_per_step'],2), 'ms/slice', round(d['tflops'],1), 'TFLOP/s')"
}
for i in 1 2; do
  for t in sycamore_m20_native sycamore_m20_fused; do
    run r3 $t ""; run r4 $t ""
  done
done
for t in sycamore_m20_native sycamore_m20_fused; do
  run r3 $t CTG_STEM_BF16X3=1; run r4 $t CTG_STEM_BF16X3=1
done
