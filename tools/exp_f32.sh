#!/bin/bash
# float32 matrix-core kernel: correctness (pairwise tests) and rates by operand memory order
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-f32}; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_pairwise.py tests/test_gpu_random_pairs.py -x -q -m gpu > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -2
for spec in "ab,bc->ac a=4096,b=2048,c=4096" "ab,bc->ac a=1048576,b=256,c=64" "ab,bc->ac a=65536,b=1024,c=512" \
            "ab,cb->ac a=8192,b=4096,c=4096" "ba,bc->ac a=4096,b=2048,c=4096" "ba,cb->ac a=4096,b=2048,c=4096"; do
  set -- $spec
  timeout 120 python tools/bench_pair.py "$1" "$2" 3 - float32 2>&1 | grep "kernel" | cut -c1-150
done 2>&1 | tee $O/rates.log
