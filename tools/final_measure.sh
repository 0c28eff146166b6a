#!/bin/bash
# End-of-round measurement on the GPU box: tests, the bench line, a kernel
# trace and the two HBM-traffic counter passes (each bounded by its own timeout).
# Outputs land in gpurun_out/final/; copy what is to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O   # (locally, delete gpurun_out/final before the call: results are merged, not mirrored)
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1
grep -E "passed|failed|error" $O/tests.log | tail -2
timeout 600 python bench.py > $O/bench.log 2>&1
tail -1 $O/bench.log > $O/bench_line.json
cut -c1-400 $O/bench_line.json
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -- $CMD > $O/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -- $CMD > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -- $CMD > $O/pmc_write.log 2>&1
cd $R
T=$(find $O/trace -name "*.db" | head -1); F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1)
python tools/rocprof_summary.py $T $O/kernels > /dev/null 2>&1; head -12 $O/kernels_kernels.txt | cut -c1-190
python tools/rocprof_summary.py $F $O/pmc_fetch_k > /dev/null 2>&1
python tools/rocprof_summary.py $W $O/pmc_write_k > /dev/null 2>&1
python tools/pmc_traffic.py $F $W 4 $O/pmc_summary.json | tail -8
find $O -name "*.db" -size +30M -delete
# per-step roofline table of one slice (HIP events on the exec's stream)
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --dump-steps $O/steps.json > /dev/null 2>&1
python tools/steps_report.py $O/steps.json 40 > $O/steps.txt 2>&1
