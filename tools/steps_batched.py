#!/usr/bin/env python
"""Per-step times of a small configuration with its slices batched (CTG_PROFILE_SLICES),
against each step's own bound max(flops / 157.3 TF, bytes / 6.1 TB/s):
  python tools/steps_batched.py C5|C3|C2 [top]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import golden_util as G  # noqa: E402
import cotengra_amd as ca  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402

which = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
if which == "C3":
    tree = ca.tree_from_record(ca.load_network(os.path.join(ROOT, "tests/golden/trees/sycamore_m10.json")))
    z = np.load(os.path.join(ROOT, "tests/golden/sycamore_m10_arrays.npz"))
    arrays = [z[f"t{i}"].astype("complex64") for i in range(tree.N)]
else:
    name = {"C2": "C2_lattice8x8_d4", "C5": "C5_hyper200"}[which]
    case = next(c for c in G.cases("tree") if c["name"] == name)
    tree = G.tree_of(case)
    arrays = [a.astype("complex64") for a in G.arrays_of(case, "complex128", tree)]
fn = HipContractor(tree)
st = fn.setup(*[torch.as_tensor(a, device="cuda") for a in arrays])
ex, plan = st["exec"], st["plan"]
nb = min(ex.batch, tree.nslices)
os.environ["CTG_PROFILE_SLICES"] = str(nb)
best = None
for _ in range(5):
    ms = ex.profile_slice(0)
    best = ms if best is None else np.minimum(best, ms)
rows = plan.describe_steps()
names = ex.step_kernels()
fl = 8.0 if plan.is_complex else 2.0
# (a step the slices of a group share runs once per group of the launch)
gs = int(plan.group_size) if getattr(ex, "batch", 1) > 1 else 1
if gs > 1:
    nb -= nb % gs
launches = [nb // gs if (gs > 1 and getattr(st_, "group", False)) else nb for st_ in plan.steps]
ideal = np.array([max(r["macs"] * fl / 157.3e12, r["bytes"] / 6.1e12) * n * 1e3 for r, n in zip(rows, launches)])
print("%s: %d slices per launch%s, sum of step times %.3f ms, sum of bounds %.3f ms" % (
    which, nb, " (slice groups of %d: shared steps once per group)" % gs if gs > 1 else "", best.sum(), ideal.sum()))
order = np.argsort(-(best - ideal))
print("%5s %-52s %9s %6s %5s %3s %9s %9s %6s %8s" % ("step", "kernel", "R", "K", "N", "Bt", "ms", "bound", "x", "TB/s"))
for i in order[:top]:
    r = rows[i]
    print("%5d %-52s %9d %6d %5d %3d %9.3f %9.3f %6.2f %8.2f" % (
        i, names[i][:52], r["R"], r["K"], r["N"], r["Bt"], best[i], ideal[i], best[i] / max(ideal[i], 1e-9),
        r["bytes"] * launches[i] / (best[i] * 1e-3) / 1e12))
fn.close()
