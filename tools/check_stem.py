#!/usr/bin/env python
"""GPU check of the fused stem kernel, case by case with diagnostics (run on the box):
fused HIP vs numpy complex128, vs the unfused HIP path, and what differs where."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cotengra_amd as ca  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402
from cotengra_amd.plan import KIND_STEM2  # noqa: E402
from oracle import contract_ref as orc  # noqa: E402

import golden_util as G  # noqa: E402
from cotengra_amd import stem  # noqa: E402

stem.gather_rate = lambda run_bytes: 5.4e12   # every pair the kernel can take, whatever the model thinks of its gathers

bad = 0
for ci, (nq, gates) in enumerate(G.STEM_CASES):
    for seed in (0, 1, 2):
        for sliced in (0, 2):
            tree = G.stem_network(nq, gates, 100 * ci + seed, sliced=sliced)
            arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=seed, dtype="complex64")
            ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
            fused = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
            plain = HipContractor(tree, fuse=False)
            plan = fused.get_plan("complex64")[0]
            shapes = [(s.stem["K1"], s.stem["N1"], s.stem["K2"], s.stem["N2"], s.stem["nr1"])
                      for s in plan.steps if s.kind == KIND_STEM2]
            try:
                got = np.asarray(fused(*arrays))
                base = np.asarray(plain(*arrays))
            except Exception as e:  # noqa: BLE001
                print(f"case {ci} seed {seed} sliced {sliced} {shapes}: EXCEPTION {e}")
                bad += 1
                continue
            scale = np.abs(ref).max()
            err = np.abs(got - ref).max() / scale
            err0 = np.abs(base - ref).max() / scale
            same = np.array_equal(got, base)
            ok = err <= max(1e-5, 8 * err0)
            print(f"case {ci} seed {seed} sliced {sliced} fused {shapes}: err {err:.2e} (unfused {err0:.2e}) "
                  f"bit-identical {same} {'ok' if ok else 'WRONG'}")
            if not ok:
                bad += 1
                d = np.abs(got - ref).reshape(-1)
                wrong = np.flatnonzero(d > 1e-4 * scale)
                print(f"   {len(wrong)} of {d.size} elements off; first {wrong[:16]}; "
                      f"nan {int(np.isnan(got).sum())}; zero {int((got == 0).sum())}")
            fused.close()
            plain.close()
print("FAILED" if bad else "ALL OK", bad)
