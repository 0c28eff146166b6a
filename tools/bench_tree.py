#!/usr/bin/env python
"""Time repeated contractions of a small golden tree on the GPU (launch-bound
regime): python tools/bench_tree.py C2_lattice8x8_d4 [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import golden_util as G  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C2_lattice8x8_d4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
case = next(c for c in G.cases("tree") if c["name"] == name)
tree = G.tree_of(case)
arrays = [a.astype("complex64") for a in G.arrays_of(case, "complex128", tree)]
fn = HipContractor(tree)
st = fn.setup(*arrays)
ex, plan = st["exec"], st["plan"]
for _ in range(3):
    ex.run_slices(0, 1, 1)
ex.sync()
t0 = time.perf_counter()
for _ in range(reps):
    ex.run_slices(0, 1, 1)
ex.sync()
dt = (time.perf_counter() - t0) / reps
print(f"{name}: {len(plan.steps)} steps, {dt*1e6:.1f} us per contraction, "
      f"{plan.flops_per_slice()/dt/1e12:.2f} TFLOP/s, hipGraph replay={'on' if os.environ.get('CTG_GRAPH') else 'off'}")
