"""The committed Sycamore m20 contraction trees (tests/golden/trees): their
recorded statistics agree with the tree accounting, and the device plan compiled
for a narrowed version of each computes what the numpy oracle computes (plan
interpreter on the CPU; the HIP run of the same trees is in test_gpu_golden)."""
import math
import os

import numpy as np
import pytest

import cotengra_amd as ca
from cotengra_amd.plan import compile_tree
from oracle import contract_ref as orc
from oracle.plan_interp import run_plan

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = ["sycamore_m20_w30.json", "sycamore_m20_w32.json", "sycamore_m20_w32_c512.json", "sycamore_m20_w32_c128.json",
     "sycamore_m20_w32_time.json", "sycamore_m20_native.json", "sycamore_m20_fused.json",
     "sycamore_m20_w33_fused.json", "sycamore_m20_w33_bf3.json", "sycamore_m20_w32_r4.json", "sycamore_m20_w32_g.json"]


def narrowed(tree, log2_width):
    tree = tree.copy()
    while tree.max_size() > 2**log2_width:
        big = max((p for p, _, _ in tree.traverse()), key=tree.get_size)
        tree.remove_ind_(next(iter(tree.get_legs(big))))
    return tree


@pytest.mark.parametrize("fixture", FIXTURES)
def test_m20_tree_fixture(fixture):
    rec = ca.load_network(os.path.join(HERE, "golden", "trees", fixture))
    tree = ca.tree_from_record(rec)
    assert tree.N == 381 and tree.output == ()
    st = rec["stats"]
    assert math.log2(tree.nslices) == pytest.approx(st["nslices_log2"])
    assert tree.contraction_cost(log=10) == pytest.approx(st["contraction_cost_log10"], abs=1e-6)
    assert tree.max_size(log=2) == pytest.approx(st["max_size_log2"])
    assert tree.contraction_cost() // tree.nslices == st["cost_per_slice"]
    # one slice of the tree narrowed to 2^10: device plan (MFMA step encoding,
    # interpreted in double precision) == the reference executor's arithmetic
    small = narrowed(tree, 10)
    arrays = ca.make_arrays_from_inputs(small.inputs, small.size_dict, seed=42, dtype="complex128", rescale=True)
    plan = compile_tree(small, "complex64")
    plan.dtype = "complex128"
    sid = small.nslices // 3
    got = run_plan(plan, arrays, slice_ids=[sid])
    ref = orc.contract_slice(small, arrays, sid)
    assert abs(complex(np.asarray(got)) - complex(ref)) <= 1e-10 * abs(complex(ref))
