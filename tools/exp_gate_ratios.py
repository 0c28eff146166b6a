#!/usr/bin/env python
"""How far is the complex64 HIP path from the complex128 oracle on the narrowed m20 trees, relative to the error numpy's
own complex64 run makes on the same slice?  (The gate of tests/test_gpu_fullwidth.py (i) is a multiple of the latter.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cotengra_amd as ca  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402
from oracle import contract_ref as orc  # noqa: E402

rel = lambda a, b: abs(complex(a) - complex(b)) / abs(complex(b))  # noqa: E731
worst = 0.0
for fixture in ["sycamore_m20_w32_c512.json", "sycamore_m20_native.json", "sycamore_m20_fused.json", "sycamore_m20_w33_bf3.json",
                "sycamore_m20_w32_r4.json", "sycamore_m20_w32_g.json"]:
    tree = ca.tree_from_record(ca.load_network(os.path.join(ROOT, "tests", "golden", "trees", fixture)))
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=42, dtype="complex64", rescale=True)
    a128 = [a.astype("complex128") for a in arrays]
    for w in (20, 24):
        small = tree.slice(target_size=2**w)
        for sid in (3, small.nslices - 1 if small.nslices < 2**62 else 12345):
            ref = orc.contract_slice(small, a128, sid)
            e_np = rel(orc.contract_slice(small, arrays, sid), ref)
            out = {}
            for mode in ("1", "0"):
                os.environ["CTG_STEM_BF16X3"] = mode
                fn = HipContractor(small)
                out[mode] = rel(fn.contract_slice(arrays, sid), ref)
                fn.close()
            del os.environ["CTG_STEM_BF16X3"]
            worst = max(worst, out["1"] / max(e_np, 1.25e-6))
            print(f"{fixture:28s} 2^{w} slice {sid}: numpy c64 {e_np:.2e}  HIP bf16x3 {out['1']:.2e} ({out['1'] / e_np:5.2f} x)  HIP fp32 {out['0']:.2e} ({out['0'] / e_np:5.2f} x)")
print("worst ratio of the default arithmetic to max(numpy's error, 1.25e-6):", round(worst, 2))
