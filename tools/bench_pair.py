#!/usr/bin/env python
"""Time single pairwise contractions on the GPU (kernel-level microbench).

  python tools/bench_pair.py "ak,kb->ab" a=32,k=1048576,b=32 [reps]
"""
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cotengra_amd as ca  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402


def main():
    eq = sys.argv[1]
    sizes = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[2].split(",")}
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    force = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] != "-" else None
    dtype = sys.argv[5] if len(sys.argv) > 5 else "complex64"
    (ta, tb), out = ca.eq_to_inputs_output(eq)
    tree = ca.ContractionTree.from_path([ta, tb], out, sizes, path=[(0, 1)])
    rng = np.random.default_rng(0)
    def make(t):
        shape = [sizes[i] for i in t]
        if int(np.prod(shape)) >= (1 << 24) and dtype == "complex64":
            import torch  # big operands: generated on the device

            return torch.view_as_complex(torch.randn(shape + [2], device="cuda", dtype=torch.float32))
        return (rng.normal(size=shape) + 1j * rng.normal(size=shape)).astype(dtype)

    arrays = [make(t) for t in (ta, tb)]
    fn = HipContractor(tree, force_kernel=force)
    st = fn.setup(*arrays)
    plan, ex = st["plan"], st["exec"]
    best = None
    for _ in range(reps):
        ms = ex.profile_slice(0)
        best = ms if best is None else np.minimum(best, ms)
    for r, m, nm in zip(plan.describe_steps(), best, ex.step_kernels()):
        if r["kind"] == "pair":
            print(f"{eq} {sizes}: {nm} R={r['R']} K={r['K']} N={r['N']} "
                  f"ms={m:.4f} TF={(8 if 'complex' in dtype else 2)*r['macs']/m/1e9:.2f} GB/s={r['bytes']/m/1e6:.0f}")
    fn.close()


if __name__ == "__main__":
    main()
