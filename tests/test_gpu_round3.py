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
