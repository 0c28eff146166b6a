"""First GPU parity tests: HIP path (through the C ABI) vs the numpy oracle on
seeded lattice networks, all four dtypes, sliced and unsliced."""
import numpy as np
import pytest

import cotengra_amd as ca
from cotengra_amd.contractor import HipContractor
from oracle import contract_ref as orc

import golden_util as G

pytestmark = pytest.mark.gpu

TOL = {"float64": 1e-11, "complex128": 1e-11}


def gate(tree, arrays, dtype, ref):
    """Double precision: 1e-11.  Single precision: max(1e-5, 8 x the error of the numpy
    oracle run in the same single precision) -- golden_util.single_gate."""
    if dtype in TOL:
        return TOL[dtype]
    return G.single_gate(ref, orc.contract(tree, arrays))


def greedy_path(inputs, output, size_dict):
    from cotengra_amd.interface import greedy_path as gp

    return gp(inputs, output, size_dict)


def lattice_tree(dims, d=2, nslice_inds=0):
    inputs, output, shapes, size_dict = ca.lattice_equation(dims, d_min=d)
    path = greedy_path(inputs, output, size_dict)
    tree = ca.ContractionTree.from_path(inputs, output, size_dict, path=path)
    # slice the indices of the largest intermediate
    if nslice_inds:
        big = max((p for p, _, _ in tree.traverse()), key=tree.get_size)
        for ix in list(tree.get_legs(big))[:nslice_inds]:
            tree.remove_ind_(ix)
    return tree


def relerr(x, ref):
    x, ref = np.asarray(x), np.asarray(ref)
    return float(np.abs(x - ref).max() / max(np.abs(ref).max(), 1e-300))


@pytest.mark.parametrize("dtype", ["float32", "float64", "complex64", "complex128"])
@pytest.mark.parametrize("nsl", [0, 2])
@pytest.mark.parametrize("force", [0, None])
def test_lattice(dtype, nsl, force):
    tree = lattice_tree([4, 4], d=3, nslice_inds=nsl)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=42, dtype=dtype, rescale=True)
    ref = orc.contract(tree, [a.astype("complex128" if "complex" in dtype else "float64") for a in arrays])
    fn = HipContractor(tree, force_kernel=force)
    out = fn(*arrays)
    assert relerr(out, ref) <= gate(tree, arrays, dtype, ref)
    fn.close()


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
def test_lattice_8x8_d4(dtype):
    tree = lattice_tree([8, 8], d=4)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=42, dtype=dtype, rescale=True)
    ref = orc.contract(tree, [a.astype("complex128") for a in arrays])
    out = HipContractor(tree)(*arrays)
    assert relerr(out, ref) <= gate(tree, arrays, dtype, ref)


def test_open_output_sliced_outer():
    # 3x3 lattice with two dangling output legs, one of them sliced (stack path)
    inputs, output, shapes, size_dict = ca.lattice_equation([3, 3], d_min=3)
    inputs = [list(t) for t in inputs]
    inputs[0].append("Y")
    inputs[8].append("Z")
    size_dict = dict(size_dict, Y=4, Z=5)
    output = ["Z", "Y"]
    path = greedy_path(inputs, output, size_dict)
    tree = ca.ContractionTree.from_path(inputs, output, size_dict, path=path)
    tree.remove_ind_("Y")
    tree.remove_ind_(inputs[4][0])
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=7, dtype="complex128")
    ref = orc.contract(tree, arrays)
    out = tree.contract(arrays)
    assert out.shape == ref.shape == (5, 4)
    assert relerr(out, ref) < 1e-11
    s3 = tree.contract_slice(arrays, 3)
    r3 = orc.contract_slice(tree, arrays, 3)
    assert relerr(s3, r3) < 1e-11
