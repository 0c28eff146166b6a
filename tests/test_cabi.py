"""CPU suite, part 2: the C-ABI shared library loads, exports every symbol
include/ctg_hip.h declares, and validates plans on the host (no GPU work)."""
import os
import re

import numpy as np
import pytest

import cotengra_amd as ca
from cotengra_amd import runtime
from cotengra_amd.plan import compile_tree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ctg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ctg_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = runtime.load()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ctg_hip.h but not exported"
    assert sorted(runtime.SYMBOLS) == syms
    assert lib.ctg_abi_version() == runtime.ABI_VERSION


def small_tree(sliced=True):
    inputs, output, shapes, size_dict = ca.lattice_equation([3, 3], d_min=2)
    tree = ca.ContractionTree.from_path(inputs, output, size_dict,
                                        path=ca.greedy_path(inputs, output, size_dict))
    if sliced:
        tree.remove_ind_(inputs[4][0])
        tree.remove_ind_(inputs[0][0])
    return tree


def test_plan_create_validate_and_query():
    tree = small_tree()
    plan = compile_tree(tree, "complex64")
    dp = runtime.DevicePlan(plan)
    assert dp.nslices == tree.nslices == 4
    ws = dp.workspace_bytes()
    assert ws["inputs"] == plan.inputs_elems * 8 and ws["arena"] == plan.arena_elems * 8
    assert ws["result"] == 8
    dp.close()


def test_plan_rejects_out_of_bounds_tables():
    tree = small_tree(sliced=False)
    plan = compile_tree(tree, "float64")
    # corrupt: shrink the arena so that some step writes outside it
    plan.arena_elems = 1
    with pytest.raises((ValueError, runtime.CtgError)) as ei:
        runtime.DevicePlan(plan)
    assert "outside" in str(ei.value) or "space" in str(ei.value)


def test_plan_rejects_bad_descriptor():
    tree = small_tree(sliced=False)
    plan = compile_tree(tree, "float64")
    ser = plan.serialise

    def broken():
        s = ser()
        s["steps"] = s["steps"].copy()
        s["steps"][0] = 7  # invalid step kind
        return s

    plan.serialise = broken
    with pytest.raises(ValueError):
        runtime.DevicePlan(plan)


def test_matrix_core_kernel_only_for_pair_steps():
    tree = ca.ContractionTree(["aab"], "b", dict(a=3, b=4))
    plan = compile_tree(tree, "float64")
    plan.steps[0].kernel = 1  # a SINGLE step cannot run on the MFMA kernels
    with pytest.raises(ValueError):
        runtime.DevicePlan(plan)


def test_no_cpu_fallback_when_library_missing(monkeypatch):
    monkeypatch.setattr(runtime, "_lib", None)
    monkeypatch.setattr(runtime, "_LIB_PATH", "/nonexistent/libctg_hip.so")
    with pytest.raises(ImportError):
        runtime.load()


def test_plain_c_driver_fixture_is_current():
    """tests/golden/cabi_plan.bin (the plan + inputs + expected result the plain-C driver
    tests/cabi_reduce.c runs) is what its generator writes for today's plan format."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(
        "make_cabi_plan", os.path.join(root, "tests", "golden", "gen", "make_cabi_plan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(os.path.join(root, "tests", "golden", "cabi_plan.bin"), "rb") as f:
        assert f.read() == mod.build()
    # the driver is C, built by build(): it must exist next to its source
    assert os.path.exists(os.path.join(root, "tests", "cabi_reduce.c"))
