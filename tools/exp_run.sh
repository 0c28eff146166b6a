#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pairwise.py tests/test_gpu_random_pairs.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --dump-steps gpurun_out/steps_w32.json > gpurun_out/exp_w32.log 2>&1
tail -1 gpurun_out/exp_w32.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms/slice %.2f TF %.1f' % (d['ms_per_step'], d['tflops']))
"
