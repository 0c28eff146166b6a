#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/exp_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/exp_tests.log | tail -3
grep -E "^FAILED|^E  " gpurun_out/exp_tests.log | head -10
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --dump-steps gpurun_out/steps_w32.json > gpurun_out/exp_w32.log 2>&1
tail -1 gpurun_out/exp_w32.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms/slice %.2f TF %.1f' % (d['ms_per_step'], d['tflops']), 'dominant', d['roofline']['kernel'], '%.1f' % d['roofline']['achieved'])
"
