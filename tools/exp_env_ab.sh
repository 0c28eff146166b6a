#!/bin/bash
# same-box comparison of environment switches on the headline tree: tools/exp_env_ab.sh "VAR=1" "VAR2=0" ... (each run twice, alternating with the default)
R=${GRAFT_REPO_ROOT:-$PWD}
run() { env $1 timeout 900 python $R/bench.py --steps 8 --warmup 2 --headline-only --no-cpu-baseline ${TREE:+--tree $R/tests/golden/trees/$TREE} 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms/slice', d['ms_per_step'], 'dominant', d['roofline']['avg_launch_ms'])"; }
for rep in 1 2; do
  run "CTG_DUMMY=0"
  for v in "$@"; do run "$v"; done
done
