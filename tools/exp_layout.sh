#!/bin/bash
# Memory path of the streaming kernel vs operand layout (contiguous rows, k slowest, mixed)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-lay}; mkdir -p $O; cd $R
for lib in peel memonly; do
  for spec in "ak,kb->ab a=134217728,k=32,b=32" "ka,kb->ab a=134217728,k=32,b=32" "xay,xyb->ab a=134217728,x=4,y=8,b=32" "axy,xyb->ab a=134217728,x=4,y=8,b=32" "ak,kb->ab a=268435456,k=16,b=16" "ak,kb->ab a=67108864,k=64,b=64"; do
    set -- $spec
    CTG_LIB=$R/cotengra_amd/lib/exp/libctg_$lib.so timeout 120 python tools/bench_pair.py "$1" "$2" 3 2>&1 | grep "kernel" | sed "s/^/$lib: /" | cut -c1-170
  done
done 2>&1 | tee $O/layout.log
timeout 60 python tools/bw_probe.py 2>&1 | tail -5 | tee -a $O/layout.log
