#!/usr/bin/env python
"""Replay single pair steps of a tree as stand-alone contractions with the
same operand layouts (operands generated on the device).

  python tools/bench_step.py tests/golden/trees/sycamore_m20_w32.json 207,206 [force] [reps]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import cotengra_amd as ca  # noqa: E402
from cotengra_amd import plan as P  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402


def main():
    rec = ca.load_network(sys.argv[1])
    steps = [int(x) for x in sys.argv[2].split(",")]
    force = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "-" else None
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    tree = ca.tree_from_record(rec)
    pl = P.compile_tree(tree, "complex64")
    dev = torch.device("cuda", 0)
    for i in steps:
        s = pl.steps[i]
        sd = {}
        for t in (s.a, s.b, s.c):
            for ix in t.inds:
                sd[ix] = tree.size_dict[ix] if ix not in tree.sliced_inds else 1
        ta, tb, out = tuple(s.a.inds), tuple(s.b.inds), tuple(s.c.inds)
        sub = ca.ContractionTree.from_path([ta, tb], out, sd, path=[(0, 1)])
        arrays = [
            torch.view_as_complex(torch.randn([sd[ix] for ix in t] + [2], device=dev, dtype=torch.float32))
            for t in (ta, tb)
        ]
        fn = HipContractor(sub, force_kernel=force)
        st = fn.setup(*arrays)
        plan, ex = st["plan"], st["exec"]
        best = None
        for _ in range(reps):
            ms = ex.profile_slice(0)
            best = ms if best is None else np.minimum(best, ms)
        names = ex.step_kernels()
        for r, m, nm in zip(plan.describe_steps(), best, names):
            if r["kind"] == "pair":
                print(f"step {i}: {nm} R={r['R']} K={r['K']} N={r['N']} ms={m:.4f} "
                      f"TF={8*r['macs']/m/1e9:.2f} GB/s={r['bytes']/m/1e6:.0f}", flush=True)
        fn.close()
        del arrays
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
