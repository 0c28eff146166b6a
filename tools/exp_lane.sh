#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-lane}; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_pairwise.py tests/test_gpu_random_pairs.py tests/test_gpu_golden.py tests/test_gpu_basic.py -x -q -m gpu > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -2
for mode in tables notables; do
  [ $mode = notables ] && export CTG_NO_LANE_TABLES=1 || unset CTG_NO_LANE_TABLES
  echo "== $mode"
  timeout 300 python tools/bench_step.py tests/golden/trees/sycamore_m20_native.json 229,175,252 - 3 2>&1 | grep "^step" | cut -c1-150
  timeout 300 python tools/bench_step.py tests/golden/trees/sycamore_m20_w32_c512.json 364,196 - 2 2>&1 | grep "^step" | cut -c1-150
done 2>&1 | tee $O/ab.log
unset CTG_NO_LANE_TABLES
for t in sycamore_m20_w32_c512 sycamore_m20_native; do
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --tree tests/golden/trees/$t.json 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['tree'], round(d['ms_per_step'],2), 'ms', round(d['tflops'],2), 'TF', {k:round(v,1) for k,v in list(d['roofline']['by_kernel_ms'].items())[:5]})"
done 2>&1 | tee -a $O/ab.log
