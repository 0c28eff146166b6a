#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cat > /tmp/rc.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
bench.host_cores = lambda: 1
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
out = bench.other_configs(dev)
print({k: (round(v["ms"], 3), v["steps_per_slice"], v["launches_per_slice"]) for k, v in out.items()})
PY
for f in 256 512 1024; do echo "== fill $f"; CTG_TILE_FILL=$f timeout 600 python /tmp/rc.py 2>&1 | tail -1; done
