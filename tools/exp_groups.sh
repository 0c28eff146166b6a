#!/bin/bash
# wave-front groups: parity test, then the small configurations with and without them
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/groups
timeout 900 python -m pytest tests/test_gpu_round2.py -q -x -m gpu -k "grouped or batching" > gpurun_out/groups/test.log 2>&1
tail -5 gpurun_out/groups/test.log
cat > /tmp/rc.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
out = bench.other_configs(dev)
print({k: (round(v["ms"], 3), v["steps_per_slice"], v["launches_per_slice"]) for k, v in out.items()})
PY
echo "== groups on"; timeout 600 python /tmp/rc.py 2>&1 | tail -1 | tee gpurun_out/groups/on.log
echo "== groups off"; CTG_NO_GROUPS=1 timeout 600 python /tmp/rc.py 2>&1 | tail -1 | tee gpurun_out/groups/off.log
echo "== no fast groups"; CTG_NO_FAST_GROUPS=1 timeout 600 python /tmp/rc.py 2>&1 | tail -1 | tee gpurun_out/groups/nofast.log
echo "== groups on again"; timeout 600 python /tmp/rc.py 2>&1 | tail -1 | tee gpurun_out/groups/on2.log
bash tools/exp_c2_trace.sh 0
