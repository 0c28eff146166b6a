#!/bin/bash
mkdir -p gpurun_out
( time timeout 900 python bench.py --tree gpurun_in/t_rs32b.json --dump-steps gpurun_out/steps_rs32b.json > gpurun_out/exp_rs32b.log 2>&1 ) 2> gpurun_out/exp_rs32b.time
tail -1 gpurun_out/exp_rs32b.log | cut -c1-3000
cat gpurun_out/exp_rs32b.time
