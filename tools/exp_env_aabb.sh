#!/bin/bash
# Same-box comparison of environment switches that is not fooled by the period-2 alternation of consecutive processes
# (every second bench process on a box runs 2-3 % slower, whatever it is asked to do: profiles/r6_process_alternation.txt):
# every setting runs twice in a row, the pairs in ABBA order.   tools/exp_env_aabb.sh "A=1" "B=1"   [TREE=...]
R=${GRAFT_REPO_ROOT:-$PWD}
run() { env $1 timeout 900 python $R/bench.py --steps 8 --warmup 2 --headline-only --no-cpu-baseline ${TREE:+--tree $R/tests/golden/trees/$TREE} 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms/slice', d['ms_per_step'], 'dominant', d['roofline']['avg_launch_ms'])"; }
A=$1; B=$2
for s in "$A" "$A" "$B" "$B" "$B" "$B" "$A" "$A"; do run "$s"; done
