#!/usr/bin/env python
"""Accuracy of the stem pairs' bf16 mode (CTG_STEM_BF16X3) on real trees: slice amplitudes of the m20
trees narrowed to width 2^24 (pairs fused from 2^12 elements so that the narrowed trees have them),
relative error against the numpy complex128 oracle -- numpy in complex64, the HIP fp32 path, the
HIP bf16 x 3 path."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CTG_FUSE_MIN_ELEMS"] = str(1 << 12)
import cotengra_amd as ca  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402
from oracle import contract_ref as orc  # noqa: E402

TREES = os.path.join(ROOT, "tests", "golden", "trees")


def rel(a, b):
    return abs(complex(a) - complex(b)) / abs(complex(b))


print("tree                      slice            pairs  |amplitude|  numpy-c64   HIP fp32    HIP bf16x3  bf16x3 / fp32")
for fixture in ("sycamore_m20_fused.json", "sycamore_m20_native.json"):
    tree = ca.tree_from_record(ca.load_network(os.path.join(TREES, fixture)))
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=42, dtype="complex64", rescale=True)
    small = tree.slice(target_size=2**24)
    a128 = [a.astype("complex128") for a in arrays]
    for sid in (3, small.nslices - 1 if small.nslices < 2**62 else 12345, 5, 77, 1000, 4242):
        ref = orc.contract_slice(small, a128, sid)
        np64 = orc.contract_slice(small, arrays, sid)
        fn = HipContractor(small)
        plan = fn.get_plan("complex64")[0]
        nf = sum(s.kind == 3 for s in plan.steps)
        os.environ.pop("CTG_STEM_BF16X3", None)
        e32 = rel(fn.contract_slice(arrays, sid), ref)
        os.environ["CTG_STEM_BF16X3"] = "1"
        names = [n for n in fn.setup(*arrays)["exec"].step_kernels() if n.startswith("stem2") and n.endswith(",true>")]
        e3 = rel(fn.contract_slice(arrays, sid), ref)
        os.environ.pop("CTG_STEM_BF16X3", None)
        fn.close()
        print(f"{fixture:25s} {sid:<16d} {nf:2d}/{len(names):<2d}  {abs(ref):.2e}    {rel(np64, ref):.2e}    {e32:.2e}    {e3:.2e}    {e3 / e32:5.2f}",
              flush=True)
