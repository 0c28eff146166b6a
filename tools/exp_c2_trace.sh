#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $R/gpurun_out/c2trace
rm -rf /tmp/c2t
timeout 600 rocprofv3 --kernel-trace -d /tmp/c2t -o c2 -- python $R/tools/run_c2.py > $R/gpurun_out/c2trace/run.log 2>&1
tail -2 $R/gpurun_out/c2trace/run.log
DB=$(find /tmp/c2t -name "*.db" | head -1)
python $R/tools/trace_timeline.py $DB ${1:-32} $R/gpurun_out/c2trace/timeline.txt | tail -40
