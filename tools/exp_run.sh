#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu > gpurun_out/exp_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/exp_tests.log | tail -3
for t in main c256 c1024; do
  if [ $t = main ]; then T=tests/golden/trees/sycamore_m20_w30.json; else T=gpurun_in/t_$t.json; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --tree $T --dump-steps gpurun_out/steps_$t.json > gpurun_out/exp_$t.log 2>&1
  tail -1 gpurun_out/exp_$t.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('$t', 'ms/slice %.2f TF %.1f' % (d['ms_per_step'], d['tflops']), 'flops/slice %.3g' % d['config']['flops_per_slice'], 'dominant', d['roofline']['kernel'], '%.1f' % d['roofline']['achieved'])
except Exception as e:
    print('$t', 'FAILED', e)
"
done
