#!/bin/bash
# FP64 / real matrix-core kernels: correctness (pairwise tests in all dtypes) and rates
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-f64}; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_pairwise.py tests/test_gpu_random_pairs.py -x -q -m gpu > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -2
for dt in complex128 float32 float64; do
  for spec in "ab,bc->ac a=4096,b=2048,c=4096" "ab,bc->ac a=1048576,b=256,c=64" "ab,bc->ac a=65536,b=1024,c=512"; do
    set -- $spec
    timeout 120 python tools/bench_pair.py "$1" "$2" 3 - $dt 2>&1 | grep "kernel" | cut -c1-150
  done
done 2>&1 | tee $O/rates.log
