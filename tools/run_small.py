#!/usr/bin/env python
"""One of the small configurations contracted repeatedly (for rocprofv3 --kernel-trace):
  python tools/run_small.py C2|C3|C5 [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import golden_util as G  # noqa: E402
import cotengra_amd as ca  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402

which = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if which == "C3":
    tree = ca.tree_from_record(ca.load_network(os.path.join(ROOT, "tests/golden/trees/sycamore_m10.json")))
    z = np.load(os.path.join(ROOT, "tests/golden/sycamore_m10_arrays.npz"))
    arrays = [z[f"t{i}"].astype("complex64") for i in range(tree.N)]
    slices = tree.nslices
else:
    name = {"C2": "C2_lattice8x8_d4", "C5": "C5_hyper200"}[which]
    case = next(c for c in G.cases("tree") if c["name"] == name)
    tree = G.tree_of(case)
    arrays = [a.astype("complex64") for a in G.arrays_of(case, "complex128", tree)]
    slices = min(tree.nslices, 64)
fn = HipContractor(tree)
st = fn.setup(*[torch.as_tensor(a, device="cuda") for a in arrays])
ex = st["exec"]
for _ in range(3):
    ex.run_slices(0, slices, 1)
ex.sync()
t0 = time.perf_counter()
for _ in range(reps):
    ex.run_slices(0, slices, 1)
ex.sync()
print(which, (time.perf_counter() - t0) / reps * 1e6, "us per", slices, "slices; steps, launches =", ex.launch_count())
fn.close()
