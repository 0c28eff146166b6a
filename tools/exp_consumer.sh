#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
CTG_PLAN_CONSUMER_ORDER=1 timeout 600 python -m pytest tests/test_gpu_golden.py -m gpu -x -q -k "tree_cases or lattice or rand" 2>&1 | grep -E "passed|failed|Error" | tail -2
cat > /tmp/rc.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
bench.host_cores = lambda: 1
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
out = bench.other_configs(dev)
print({k: (round(v["ms"], 3), v["steps_per_slice"], v["launches_per_slice"]) for k, v in out.items()})
PY
for m in 0 1 0 1; do echo "== consumer order $m"; CTG_PLAN_CONSUMER_ORDER=$m timeout 300 python /tmp/rc.py 2>&1 | tail -1; done
