"""GPU suite, round 3.

* ``contract(order=f)`` follows the reference's ordered traversal and gives the very bits of
  the default order (the schedule changes lifetimes, not values);
"""
import json
import os

import numpy as np
import pytest

import cotengra_amd as ca
from oracle import contract_ref as orc

import golden_util as G
from test_host_round3 import ORDERS, TRAVERSE, traverse_tree

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["lattice4x4_s0", "hyper24_s0", "randreg30_s4", "lattice3x3x3_s2"])
@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
def test_ordered_contract_same_bits(name, dtype):
    case = next(c for c in TRAVERSE if c["name"] == name)
    tree = traverse_tree(case)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=3, dtype=dtype, rescale=True)
    ref = orc.contract(tree, [a.astype("complex128") for a in arrays])
    base = np.asarray(tree.contract(arrays))
    tol = 1e-10 if dtype == "complex128" else G.single_gate(ref, orc.contract(tree, arrays))
    assert G.relerr(base, ref) <= tol
    for oname, make in ORDERS.items():
        got = np.asarray(tree.contract(arrays, order=make(tree)))
        assert got.dtype == base.dtype and np.array_equal(got, base), oname


# ---------------------------------------------------------------------- #
# fused stem pairs: csrc/ctg_stem.hip against the oracle and the unfused path
# ---------------------------------------------------------------------- #

from cotengra_amd.contractor import HipContractor  # noqa: E402
from cotengra_amd.plan import KIND_STEM2  # noqa: E402


@pytest.mark.parametrize("sliced", [0, 2])
@pytest.mark.parametrize("case", range(len(G.STEM_CASES)))
def test_fused_stem_pairs(case, sliced):
    nq, gates = G.STEM_CASES[case]
    tree = G.stem_network(nq, gates, 100 * case, sliced=sliced)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=case, dtype="complex64")
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    fused = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    plain = HipContractor(tree, fuse=False)
    plan = fused.get_plan("complex64")[0]
    n_fused = sum(s.kind == KIND_STEM2 for s in plan.steps)
    got = np.asarray(fused(*arrays))
    base = np.asarray(plain(*arrays))
    if n_fused:
        st = fused.setup(*arrays)
        assert sum(n.startswith("stem2_kernel") for n in st["exec"].step_kernels()) == n_fused
    gate = G.single_gate(ref, orc.contract(tree, arrays))
    assert G.relerr(got, ref) <= gate, (G.relerr(got, ref), gate)
    assert G.relerr(base, ref) <= gate
    # strip_exponent: one factor per fused pair, same number
    m, e = fused(*arrays, strip_exponent=True)
    assert G.relerr(np.asarray(m).astype("complex128") * 10.0**e, ref) <= gate
    # slice by slice = all slices in one batched run
    if tree.nslices > 1:
        parts = sum(np.asarray(fused.contract_slice(arrays, i)) for i in range(tree.nslices))
        assert G.relerr(parts, ref) <= gate
    fused.close()
    plain.close()
