#!/bin/bash
# Knock-out builds of the float32 matrix-core kernel (tools/build_variants.py rk*=-DCTG_REAL_KO_...)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for lib in cotengra_amd/lib/libctg_hip.so cotengra_amd/lib/exp/libctg_rk*.so; do
  n=$(basename $lib .so); echo "== ${n#libctg_}"
  for spec in "ab,cb->ac a=8192,b=4096,c=4096" "ab,bc->ac a=8192,b=4096,c=4096"; do
    set -- $spec
    CTG_LIB=$R/$lib timeout 120 python tools/bench_pair.py "$1" "$2" 3 - float32 2>&1 | grep "kernel" | cut -c1-150
  done
done
