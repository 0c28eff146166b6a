#!/usr/bin/env python
"""Per-step timings of one contraction of a golden tree case:
python tools/profile_tree.py C2_lattice8x8_d4 [top]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as G  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C2_lattice8x8_d4"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
case = next(c for c in G.cases("tree") if c["name"] == name)
tree = G.tree_of(case)
arrays = [a.astype("complex64") for a in G.arrays_of(case, "complex128", tree)]
fn = HipContractor(tree)
st = fn.setup(*arrays)
ex, plan = st["exec"], st["plan"]
best = None
for _ in range(5):
    ms = ex.profile_slice(0)
    best = ms if best is None else np.minimum(best, ms)
rows = plan.describe_steps()
names = ex.step_kernels()
print(f"{name}: {len(rows)} steps, sum of step times {best.sum()*1e3:.1f} us")
order = np.argsort(-best)
for i in order[:top]:
    r = rows[i]
    print(f"  #{i:3d} {names[i]:44s} {r['label'][:34]:34s} {best[i]*1e3:8.1f} us  "
          f"{8*r['macs']/best[i]/1e9 if best[i] > 0 else 0:7.2f} TF")
