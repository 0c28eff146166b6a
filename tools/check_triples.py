#!/usr/bin/env python
"""GPU check of the three-step tiles (CTG_STEM_TRIPLES, stem.build_stem_triple; round 4): the 48 random
stems of the test suite planned with every triple the library has a kernel for, in both arithmetics,
against numpy complex128 and against the unfused HIP path, with diagnostics (run on the box).

    python tools/check_triples.py [first seed = 0] [n = 48]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CTG_STEM_TRIPLES"] = "1"

import cotengra_amd as ca  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402
from cotengra_amd.plan import KIND_STEM2  # noqa: E402
from oracle import contract_ref as orc  # noqa: E402

import golden_util as G  # noqa: E402
from cotengra_amd import stem  # noqa: E402

stem.gather_rate = lambda run_bytes: 5.4e12   # everything the kernel can take, whatever the model thinks of it
stem.TRIPLE_STAGE_RATE = {n: 1e15 for n in stem.TRIPLE_STAGE_RATE}
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 48
bad = n_tri = 0
for seed in range(first, first + count):
    tree = G.random_stem(seed)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=seed, dtype="complex64")
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    plain = HipContractor(tree, fuse=False)
    base = np.asarray(plain(*arrays))
    plain.close()
    scale = np.abs(ref).max()
    err0 = np.abs(base - ref).max() / scale
    for mode in (True, False):
        # (triples are planned for the bf16 x 3 arithmetic; the fp32 kernels of the same shapes serve an
        # executor whose arithmetic is switched afterwards)
        fused = HipContractor(tree, fuse=True, fuse_min_elems=1 << 9, stem_bf16x3=True)
        if not mode:
            fused.get_plan("complex64")
            fused.stem_bf16x3 = False
        plan = fused.get_plan("complex64")[0]
        tri = [s.label.split(" rows")[0] for s in plan.steps if s.kind == KIND_STEM2 and s.stem.get("KM")]
        if not tri:
            fused.close()
            continue
        n_tri += len(tri)
        try:
            got = np.asarray(fused(*arrays))
            names = [n for n in fused.setup(*arrays)["exec"].step_kernels() if n.startswith("stem2_kernel")]
        except Exception as e:  # noqa: BLE001
            print(f"seed {seed} bf16x3 {mode} {tri}: EXCEPTION {e}")
            bad += 1
            fused.close()
            continue
        err = np.abs(got - ref).max() / scale
        ok = err <= max(1e-5, 8 * err0)
        print(f"seed {seed} bf16x3 {mode} {tri} {[n for n in names if n.count(',') > 11]}: err {err:.2e} (unfused {err0:.2e}) {'ok' if ok else 'WRONG'}")
        if not ok:
            bad += 1
            d = np.abs(got - ref).reshape(-1)
            wrong = np.flatnonzero(d > 1e-4 * scale)
            print(f"   {len(wrong)} of {d.size} elements off; first {wrong[:16]}; nan {int(np.isnan(got).sum())}; zero {int((got == 0).sum())}")
        fused.close()
print("FAILED" if bad else "ALL OK", bad, "triples run:", n_tri)
