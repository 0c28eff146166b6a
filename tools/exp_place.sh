#!/bin/bash
# the arena placement probe (ctg_runtime.hip: place_arena) over consecutive bench processes: tools/exp_place.sh [n] [TREE=...]
R=${GRAFT_REPO_ROOT:-$PWD}
for i in $(seq 1 ${1:-6}); do
  CTG_ARENA_DEBUG=1 timeout 900 python $R/bench.py --steps 8 --warmup 2 --headline-only --no-cpu-baseline ${TREE:+--tree $R/tests/golden/trees/$TREE} 2> /tmp/place.err | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/slice', d['ms_per_step'], 'dominant', d['roofline']['avg_launch_ms'])"
  grep "arena placement" /tmp/place.err | tail -1
done
