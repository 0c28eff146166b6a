#!/usr/bin/env python
"""The unsliced 8x8 lattice (C2) contracted repeatedly (for rocprofv3 --kernel-trace)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import golden_util as G  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402

case = next(c for c in G.cases("tree") if c["name"] == "C2_lattice8x8_d4")
tree = G.tree_of(case)
arrays = [a.astype("complex64") for a in G.arrays_of(case, "complex128", tree)]
fn = HipContractor(tree)
st = fn.setup(*[torch.as_tensor(a, device="cuda") for a in arrays])
ex = st["exec"]
for _ in range(5):
    ex.run_slices(0, 1, 1)
ex.sync()
t0 = time.perf_counter()
for _ in range(100):
    ex.run_slices(0, 1, 1)
ex.sync()
print("C2:", (time.perf_counter() - t0) / 100 * 1e6, "us per contraction")
fn.close()
