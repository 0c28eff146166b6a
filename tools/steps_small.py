import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import golden_util as G
from cotengra_amd.contractor import HipContractor
for name in ("C2_lattice8x8_d4",):
    case = next(c for c in G.cases("tree") if c["name"] == name)
    tree = G.tree_of(case)
    arrays = [a.astype("complex64") for a in G.arrays_of(case, "complex128", tree)]
    fn = HipContractor(tree); st = fn.setup(*arrays); ex, plan = st["exec"], st["plan"]
    best=None
    for _ in range(5):
        ms = ex.profile_slice(0); best = ms if best is None else np.minimum(best, ms)
    rows = plan.describe_steps(); names = ex.step_kernels()
    print(name, "sum of step times (event-bracketed) %.1f us" % (best.sum()*1e3))
    order = np.argsort(-best)
    for i in order[:12]:
        r=rows[i]; print("  step %3d %-44s R=%-7d K=%-6d N=%-5d Bt=%d  %.1f us" % (i, names[i], r['R'], r['K'], r['N'], r['Bt'], best[i]*1e3))
    print("  steps under 8 us:", int((best<0.008).sum()), "sum %.1f us" % (best[best<0.008].sum()*1e3))
