#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests/test_gpu_golden.py tests/test_gpu_round2.py tests/test_gpu_random_pairs.py tests/test_gpu_pairwise.py -m gpu -x -q -n 4 2>&1 | tail -3
for m in 0 1; do echo "== CTG_ROWWISE=$m"; CTG_ROWWISE=$m python tools/steps_batched.py C5 ${1:-14} 2>&1 | grep -v amdgpu.ids; done
bash tools/exp_fill.sh 2>&1 | tail -1
