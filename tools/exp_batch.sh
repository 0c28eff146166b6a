#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-batch}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_golden.py tests/test_circuits.py tests/test_gpu_basic.py -x -q -m gpu > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -2; grep -B5 -A25 "Error\|FAILED" $O/tests.log | head -60
timeout 600 python bench.py --steps 2 --warmup 1 > $O/bench.log 2>&1; tail -1 $O/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('headline', round(d['ms_per_step'],1), 'ms', round(d['tflops'],1), 'TF')
for k,v in d['configs'].items(): print(k, {x: (round(v[x],3) if isinstance(v[x],float) else v[x]) for x in ('ms','slices_per_sec','tflops','mixed_roofline_frac','speedup_vs_cpu_oracle')})
"
