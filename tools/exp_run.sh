#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pairwise.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/exp_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/exp_tests.log | tail -3
CTG_PERSIST_NK=100000 timeout 600 python -m pytest tests/test_gpu_pairwise.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/exp_tests_p.log 2>&1
grep -E "passed|failed|error" gpurun_out/exp_tests_p.log | tail -3
for nk in 0 4 8 32 100000; do
  CTG_PERSIST_NK=$nk timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --tree gpurun_in/t_rs32.json --dump-steps gpurun_out/steps_rs32_p$nk.json > gpurun_out/exp_p$nk.log 2>&1
  tail -1 gpurun_out/exp_p$nk.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('persist nk<=$nk', 'ms/slice %.2f TF %.1f' % (d['ms_per_step'], d['tflops']))
except Exception as e:
    print('$nk', 'FAILED', e)
"
done
