#!/bin/bash
# Round-2 GPU check: whole GPU suite, the bench line (plain and under a launcher),
# RCCL two-ranks-one-device probe.  Logs under gpurun_out/$1.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-r2a}; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > $O/tests.log 2>&1; tail -25 $O/tests.log
timeout 90 python tools/exp_rccl_same_gpu.py > $O/rccl_same_gpu.log 2>&1; tail -5 $O/rccl_same_gpu.log
timeout 900 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_line.json; cut -c1-600 $O/bench_line.json; echo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_launcher.log 2>&1; tail -1 $O/bench_launcher.log | cut -c1-300; echo
timeout 120 python bench.py --gpus 2 --steps 2 > $O/bench_gpus2.log 2>&1; tail -3 $O/bench_gpus2.log | cut -c1-300
