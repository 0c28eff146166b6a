#!/bin/bash
# The HBM-traffic counter passes of tools/final_measure.sh alone (FETCH_SIZE / WRITE_SIZE in
# separate rocprofv3 --pmc runs), for the benchmark tree and the time-to-solution tree.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
mkdir -p $O
for tree in sycamore_m20_w32_c512 sycamore_m20_native; do
  CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --tree $R/tests/golden/trees/$tree.json"
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/pmc_fetch_$tree $O/pmc_write_$tree
  timeout ${1:-60} rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$tree -- $CMD > $O/pmc_fetch_$tree.log 2>&1
  timeout ${1:-60} rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_$tree -- $CMD > $O/pmc_write_$tree.log 2>&1
  cd $R
  F=$(find $O/pmc_fetch_$tree -name "*.db" | head -1); W=$(find $O/pmc_write_$tree -name "*.db" | head -1)
  python tools/pmc_traffic.py $F $W 4 $O/pmc_summary_$tree.json $tree.json | tail -4
done
find $O -name "*.db" -delete
find $O -type d -empty -delete
