#!/bin/bash
# Round 4: one slice (3 timed) of each candidate tree given on the command line (paths relative to the repo),
# default arithmetic, and the per-step table of each; then the 8x8 lattice with several tile-fill thresholds.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4_trees; rm -rf $O; mkdir -p $O
cd $R
for t in "$@"; do
  n=$(basename $t .json)
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --headline-only --tree $R/$t --dump-steps $O/steps_$n.json > $O/$n.log 2>&1
  python - <<PY
import json
try:
    d = json.loads(open("$O/$n.log").read().strip().splitlines()[-1])
    print("$n", "%.1f ms/slice" % d["ms_per_step"], "2^%.0f slices" % d["config"]["nslices_log2"], "%.3e s" % d["est_time_total_s"], "%.1f TF" % d["tflops"])
except Exception as e:
    print("$n FAILED", e)
PY
  python tools/steps_report.py $O/steps_$n.json 30 > $O/steps_$n.txt 2>&1
done 2>&1 | tee $O/summary.txt
for f in 128 256 512 1024; do
  echo "CTG_TILE_FILL=$f $(CTG_TILE_FILL=$f timeout 120 python tools/run_c2.py 2>&1 | tail -1)"
done 2>&1 | tee -a $O/summary.txt
