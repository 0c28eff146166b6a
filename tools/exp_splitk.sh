#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests/test_gpu_golden.py tests/test_gpu_round2.py tests/test_circuits.py -m gpu -x -q -n 4 2>&1 | tail -3
cat > /tmp/rc.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
bench.host_cores = lambda: 1
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
out = bench.other_configs(dev)
print({k: (round(v["ms"], 3), v["steps_per_slice"], v["launches_per_slice"]) for k, v in out.items()})
PY
echo "== split per plan (new)"; timeout 600 python /tmp/rc.py 2>&1 | tail -1
echo "== split per slice (old)"; CTG_SPLITK_PER_SLICE=1 timeout 600 python /tmp/rc.py 2>&1 | tail -1
echo "== single-slice loop, new / old"
cat > /tmp/one.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import cotengra_amd as ca
from cotengra_amd.contractor import HipContractor
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
tree = ca.tree_from_record(ca.load_network(os.path.join(ROOT, "tests/golden/trees/sycamore_m10.json")))
z = np.load(os.path.join(ROOT, "tests/golden/sycamore_m10_arrays.npz"))
arrays = [z[f"t{i}"].astype("complex64") for i in range(tree.N)]
fn = HipContractor(tree); st = fn.setup(*[torch.as_tensor(a, device="cuda") for a in arrays]); ex = st["exec"]
for i in range(64): ex.run_slices(i, 1, 1)
ex.sync(); t0 = time.perf_counter()
for r in range(5):
    for i in range(64): ex.run_slices(i, 1, 1)
ex.sync(); print("m10 slice by slice: %.1f us per slice" % ((time.perf_counter() - t0) / 320 * 1e6))
PY
python /tmp/one.py 2>&1 | tail -1; CTG_SPLITK_PER_SLICE=1 python /tmp/one.py 2>&1 | tail -1
