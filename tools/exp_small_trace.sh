#!/bin/bash
# kernel timeline of one repetition of a small configuration: exp_small_trace.sh C5
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
for W in "$@"; do
mkdir -p $R/gpurun_out/trace_$W
rm -rf /tmp/tr_$W
timeout 600 rocprofv3 --kernel-trace -d /tmp/tr_$W -o t -- python $R/tools/run_small.py $W 10 > $R/gpurun_out/trace_$W/run.log 2>&1
grep "us per" $R/gpurun_out/trace_$W/run.log
DB=$(find /tmp/tr_$W -name "*.db" | head -1)
python $R/tools/trace_timeline.py $DB 0 $R/gpurun_out/trace_$W/timeline.txt > /dev/null
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/trace_$W/sum | head -14
done
