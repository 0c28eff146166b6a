#!/bin/bash
# Knock-out experiment for the fused stem kernel: one slice of the native m20 tree per
# variant library (tools/build_variants.py ko*=-DCTG_STEM_KO_...), fused steps only.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-stemko}; mkdir -p $O; cd $R
T=tests/golden/trees/sycamore_m20_native.json
for lib in cotengra_amd/lib/libctg_hip.so cotengra_amd/lib/exp/libctg_sk*.so; do
  n=$(basename $lib .so); n=${n#libctg_}
  CTG_LIB=$R/$lib timeout 200 python bench.py --tree $T --headline-only --no-cpu-baseline --steps 2 \
     --dump-steps $O/steps_$n.json > $O/bench_$n.log 2>&1
  echo "== $n"; python tools/steps_report.py $O/steps_$n.json 60 | grep -E "^total|stem2" | grep -v "^  " | cut -c1-120 | head -14
done
