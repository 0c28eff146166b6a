"""CPU suite, part 1: the numpy oracle and the host planner against the golden
vectors produced by the real reference (tests/golden/gen/make_golden.py).

* the oracle's results equal the reference's stored outputs (1e-11);
* our tree class reproduces the reference's linear IR `extract_contractions`
  exactly (host-planner parity, pure metadata);
* the compiled device plan, run through the numpy plan interpreter (same
  addressing semantics as the HIP kernels), equals the reference's outputs --
  this validates offset tables / arena / slice offsets without a GPU.
"""
import numpy as np
import pytest

from cotengra_amd.plan import compile_tree
from oracle import contract_ref as orc
from oracle.plan_interp import run_plan

import golden_util as G

TREE_CASES = G.cases("tree")
EQ_CASES = G.cases("eq")
SLOW = {"C5_hyper200", "C4_m20_w30_narrow20"}


def ir_repr(ops):
    return sorted(repr(tuple(op)) for op in ops)


@pytest.mark.parametrize("case", TREE_CASES, ids=[c["name"] for c in TREE_CASES])
def test_tree_ir_matches_reference(case):
    tree = G.tree_of(case)
    assert ir_repr(orc.extract_contractions(tree)) == case["ir"]
    st = case["stats"]
    if st["nslices"] < 2**62:
        assert tree.nslices == st["nslices"]
    if tree.N > 1:
        assert tree.contraction_cost() // tree.nslices == st["cost_per_slice"]
        assert tree.max_size() == st["max_size"]
        assert tree.peak_size() == st["peak_size"]
        # path round trip
        assert [list(p) for p in tree.get_path()] == case["path"]


@pytest.mark.parametrize("case", TREE_CASES, ids=[c["name"] for c in TREE_CASES])
def test_oracle_matches_reference_outputs(case):
    tree = G.tree_of(case)
    for dt in case["dtypes"]:
        arrays = G.arrays_of(case, dt, tree)
        if case["slice_ids"]:
            ids = case["slice_ids"] if case["name"] not in SLOW else case["slice_ids"][:2]
            for i in ids:
                got = orc.contract_slice(tree, arrays, i)
                assert G.relerr(got, G.expected(f"{case['name']}/{dt}/slice{i}")) < 1e-11
        else:
            got = orc.contract(tree, arrays)
            assert G.relerr(got, G.expected(f"{case['name']}/{dt}")) < 1e-11


@pytest.mark.parametrize("case", EQ_CASES, ids=[c["name"] for c in EQ_CASES])
def test_oracle_matches_reference_equations(case):
    for dt in ("complex128", "float64"):
        tree, arrays = G.eq_tree_and_arrays(case, dt)
        ref = G.expected(f"{case['name']}/{dt}")
        assert G.relerr(orc.contract(tree, arrays), ref) < 1e-11
        assert G.relerr(np.einsum(case["eq"], *arrays), ref) < 1e-10


@pytest.mark.parametrize("case", TREE_CASES, ids=[c["name"] for c in TREE_CASES])
def test_device_plan_semantics_on_cpu(case):
    if case["name"] in SLOW or case["stats"]["max_size"] > 1 << 16:
        pytest.skip("plan interpreter is for small cases")
    tree = G.tree_of(case)
    dt = case["dtypes"][0]
    arrays = G.arrays_of(case, dt, tree)
    for force in (None, 0):
        plan = compile_tree(tree, dt, force_kernel=force)
        if case["slice_ids"]:
            continue
        got = run_plan(plan, arrays)
        assert G.relerr(got, G.expected(f"{case['name']}/{dt}")) < 1e-11


@pytest.mark.parametrize("case", EQ_CASES, ids=[c["name"] for c in EQ_CASES])
def test_device_plan_semantics_equations(case):
    tree, arrays = G.eq_tree_and_arrays(case, "complex128")
    plan = compile_tree(tree, "complex128")
    got = run_plan(plan, arrays)
    assert G.relerr(got, G.expected(f"{case['name']}/complex128")) < 1e-11


def test_mfma_plan_semantics_on_cpu():
    """complex64 plans choose the MFMA step encoding (separate batch tables);
    check that encoding too, on the lattice where such steps appear."""
    case = next(c for c in TREE_CASES if c["name"] == "C2_lattice8x8_d4")
    tree = G.tree_of(case)
    arrays = G.arrays_of(case, "complex128", tree)
    plan = compile_tree(tree, "complex64")
    assert any(s.kernel == 1 for s in plan.steps)
    plan.dtype = "complex128"  # interpret the same tables in double precision
    got = run_plan(plan, arrays)
    assert G.relerr(got, G.expected("C2_lattice8x8_d4/complex128")) < 1e-11
