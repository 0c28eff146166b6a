#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for m in 1 2 4 16 64; do echo "== oversub $m"; CTG_STREAM_OVERSUB=$m timeout 300 python tools/bench_step.py tests/golden/trees/sycamore_m20_native.json ${1:-243,237,232} - 3 2>&1 | grep "^step" | cut -c1-150; done
