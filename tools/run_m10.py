#!/usr/bin/env python
"""All 64 slices of the Sycamore m10 amplitude, a few times (for rocprofv3 --stats)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cotengra_amd as ca  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "m10"
if which == "m10":
    tree = ca.tree_from_record(ca.load_network(os.path.join(ROOT, "tests", "golden", "trees", "sycamore_m10.json")))
    z = np.load(os.path.join(ROOT, "tests", "golden", "sycamore_m10_arrays.npz"))
    arrays = [z[f"t{i}"].astype("complex64") for i in range(tree.N)]
    n = tree.nslices
else:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_util as G

    case = next(c for c in G.cases("tree") if c["name"] == "C5_hyper200")
    tree = G.tree_of(case)
    arrays = [a.astype("complex64") for a in G.arrays_of(case, "complex128", tree)]
    n = 64
fn = HipContractor(tree)
st = fn.setup(*[torch.as_tensor(a, device="cuda") for a in arrays])
ex = st["exec"]
ex.zero_result()
ex.run_slices(0, n, 1)
ex.sync()
t0 = time.perf_counter()
for _ in range(5):
    ex.run_slices(0, n, 1)
ex.sync()
print(which, n, "slices:", (time.perf_counter() - t0) / 5 * 1e3, "ms")
fn.close()
