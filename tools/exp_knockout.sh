#!/bin/bash
# Knock-out experiment: replay representative steps of the native m20 tree with
# every variant library under cotengra_amd/lib/exp (tools/build_variants.py).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-ko}; mkdir -p $O; cd $R
STEPS=${2:-229,243,237,232,152,252}
for lib in cotengra_amd/lib/exp/libctg_*.so; do
  n=$(basename $lib .so); n=${n#libctg_}
  CTG_LIB=$R/$lib timeout 300 python tools/bench_step.py tests/golden/trees/sycamore_m20_native.json $STEPS - 3 > $O/$n.log 2>&1
  echo "== $n"; grep "^step" $O/$n.log | cut -c1-150
done
