"""CPU suite: the native greedy path finder and slicer (csrc/ctg_pathfind.cpp)
against numbers frozen from the reference's ``optimize_greedy`` and
``ContractionTree.slice`` (tests/golden/gen/make_pathfind.py)."""
import json
import math
import os

import numpy as np
import pytest

import cotengra_amd as ca
from cotengra_amd import pathfind
from oracle import contract_ref as orc

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "pathfind_cases.json"), encoding="utf-8"))["cases"]


def net(case):
    return [tuple(t) for t in case["inputs"]], tuple(case["output"]), case["size_dict"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_greedy_path_is_identical_to_the_reference(case):
    """Integer work is held to exact agreement: at temperature 0 the native
    finder (scores, candidate generation, tie-breaking) must take the same
    pair at every one of the N - 1 steps as the reference's ``optimize_greedy``
    did (path frozen by tests/golden/gen/make_pathfind.py)."""
    inputs, output, size_dict = net(case)
    ssa = pathfind.greedy_ssa_path(inputs, output, size_dict)
    assert [sorted(p) for p in ssa] == case["ref_greedy_ssa_path"]
    tree = pathfind.greedy_tree(inputs, output, size_dict)
    assert tree.is_complete()
    assert abs(tree.contraction_cost(log=10) - case["ref_greedy_log10_flops"]) < 1e-9
    assert tree.max_size(log=2) == case["ref_greedy_log2_width"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_slicer_reaches_target_with_reference_like_overhead(case):
    inputs, output, size_dict = net(case)
    tree = pathfind.greedy_tree(inputs, output, size_dict)
    sliced = tree.slice(target_size=case["slice_target"])
    assert sliced.max_size() <= case["slice_target"]
    assert tree.sliced_inds == {} and len(sliced.sliced_inds) >= 1
    # total work within 2x of what the reference's SliceFinder reaches
    assert sliced.contraction_cost(log=10) <= case["ref_sliced_log10_flops"] + math.log10(2.0)
    # removing indices never changes the value: checked on the small cases below


def test_paths_are_valid_and_values_agree():
    inputs, output, shapes, size_dict = ca.lattice_equation([3, 4], d_min=2, d_max=3, seed=5)
    arrays = ca.make_arrays_from_inputs(inputs, size_dict, seed=1, dtype="complex128")
    eq = ca.inputs_output_to_eq(inputs, output)
    ref = np.einsum(eq, *arrays, optimize=True)
    for kwargs in ({}, {"costmod": 0.5}, {"temperature": 0.3, "seed": 7}, {"temperature": 1.0, "seed": 8}):
        ssa = pathfind.greedy_ssa_path(inputs, output, size_dict, **kwargs)
        assert len(ssa) == len(inputs) - 1
        used = [x for pair in ssa for x in pair]
        assert sorted(used) == list(range(2 * len(inputs) - 2))  # every id consumed exactly once
        tree = ca.ContractionTree.from_path(inputs, output, size_dict, ssa_path=ssa)
        assert np.allclose(orc.contract(tree, arrays), ref, rtol=1e-10, atol=1e-12)
    tree = pathfind.random_greedy_tree(inputs, output, size_dict, repeats=8, minimize="combo-64")
    sliced = pathfind.slice_tree(tree, target_size=max(tree.max_size() // 8, 2))
    assert sliced.nslices > 1 and sliced.max_size() <= max(tree.max_size() // 8, 2)
    assert np.allclose(orc.contract(sliced, arrays), ref, rtol=1e-10, atol=1e-12)
    # target_slices and the linear-path convention
    assert tree.slice(target_slices=4).nslices >= 4
    lin = pathfind.greedy_path(inputs, output, size_dict)
    t2 = ca.ContractionTree.from_path(inputs, output, size_dict, path=lin)
    assert np.allclose(orc.contract(t2, arrays), ref, rtol=1e-10, atol=1e-12)


def test_hyper_and_disconnected_networks():
    # hyper index 'h' on three tensors + output, and a disconnected scalar pair
    inputs = [("a", "h"), ("h", "b"), ("h", "c"), ("x",), ("x",)]
    output = ("a", "b", "c", "h")
    size_dict = dict(a=2, b=3, c=2, h=4, x=5)
    arrays = ca.make_arrays_from_inputs(inputs, size_dict, seed=2, dtype="float64")
    ref = np.einsum("ah,hb,hc,x,x->abch", *arrays)
    tree = pathfind.greedy_tree(inputs, output, size_dict)
    assert np.allclose(orc.contract(tree, arrays), ref)
    sliced = tree.slice(target_size=8, allow_outer=True)
    assert sliced.max_size() <= 8
    assert np.allclose(orc.contract(sliced, arrays), ref)
    with pytest.raises(ValueError):
        tree.slice(target_size=1, allow_outer=False)   # only output indices could do that


def test_errors():
    with pytest.raises(ValueError):
        pathfind.greedy_ssa_path([("a", "b"), ("b", "c")], ("a", "c"), dict(a=2, b=2, c=2), costmod=0.0)
    with pytest.raises(ValueError):
        ca.ContractionTree.from_path([("a",), ("a",)], (), dict(a=2), path=[(0, 1)]).slice()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_subtree_reconfigure_matches_reference_quality(case):
    inputs, output, size_dict = net(case)
    tree = pathfind.greedy_tree(inputs, output, size_dict)
    rf = pathfind.subtree_reconfigure(tree, subtree_size=8, minimize="flops")
    assert rf.is_complete() and rf.contraction_cost() <= tree.contraction_cost()
    assert rf.contraction_cost(log=10) <= case["ref_reconf8_flops_log10_flops"] + 0.3
    rc = pathfind.subtree_reconfigure(tree, subtree_size=8, minimize="combo-256")
    ours = math.log10(rc.contraction_cost() + 256 * rc.total_write())
    base = math.log10(tree.contraction_cost() + 256 * tree.total_write())
    assert ours <= base + 1e-9 and ours <= case["ref_reconf8_combo256_log10_cost"] + 0.3


def test_reconfigure_preserves_value_and_slicing():
    inputs, output, shapes, size_dict = ca.lattice_equation([4, 4], d_min=2, d_max=3, seed=5)
    arrays = ca.make_arrays_from_inputs(inputs, size_dict, seed=1, dtype="complex128")
    ref = np.einsum(ca.inputs_output_to_eq(inputs, output), *arrays, optimize=True)
    tree = pathfind.greedy_tree(inputs, output, size_dict, temperature=1.0, seed=3)
    for ss, mm in ((3, "flops"), (8, "flops"), (11, "combo-64"), (9, "combo")):
        t2 = pathfind.subtree_reconfigure(tree, subtree_size=ss, minimize=mm)
        assert np.allclose(orc.contract(t2, arrays), ref, rtol=1e-10, atol=1e-12)
    # a sliced tree keeps its sliced indices; the per-slice cost is what goes down
    sliced = tree.slice(target_size=max(tree.max_size() // 4, 2))
    t3 = pathfind.subtree_reconfigure(sliced, subtree_size=8)
    assert set(t3.sliced_inds) == set(sliced.sliced_inds)
    assert t3.contraction_cost() <= sliced.contraction_cost()
    assert np.allclose(orc.contract(t3, arrays), ref, rtol=1e-10, atol=1e-12)
    # interleaved slicing + reconfiguration reaches the target
    t4 = pathfind.slice_and_reconfigure(tree, target_size=max(tree.max_size() // 8, 2), minimize="combo-64")
    assert t4.max_size() <= max(tree.max_size() // 8, 2) and t4.nslices <= 4096
    assert np.allclose(orc.contract(t4, arrays), ref, rtol=1e-10, atol=1e-12)
    with pytest.raises(ValueError):
        pathfind.subtree_reconfigure(tree, subtree_size=17)
    with pytest.raises(ValueError):
        pathfind.subtree_reconfigure(tree, minimize="size")


def modelled_seconds(tree, model):
    """Sum over the tree's contractions of the machine model's step price (the
    Python restatement of CostModel in csrc/ctg_pathfind.cpp)."""
    total = 0.0
    for p, l, r in tree.traverse():
        ll, rl, pl = tree.get_legs(l), tree.get_legs(r), tree.get_legs(p)
        size = lambda legs: math.prod(tree.size_dict[ix] for ix in legs if ix not in tree.sliced_inds)
        k = math.prod(tree.size_dict[ix] for ix in ll if ix in rl and ix not in pl and ix not in tree.sliced_inds)
        keep_l = math.prod(tree.size_dict[ix] for ix in ll if ix in pl and ix not in rl and ix not in tree.sliced_inds)
        keep_r = math.prod(tree.size_dict[ix] for ix in rl if ix in pl and ix not in ll and ix not in tree.sliced_inds)
        macs = math.prod(tree.size_dict[ix] for ix in set(ll) | set(rl) if ix not in tree.sliced_inds)
        total += model.step_seconds(macs, size(ll) + size(rl) + size(pl), k, min(keep_l, keep_r))
    return total


def test_reconfigure_under_a_machine_model():
    """minimize="time" / a MachineModel: the native search lowers the modelled
    seconds (never raises them), keeps value and slicing; the model's two regimes
    pull in different directions."""
    inputs, output, shapes, size_dict = ca.lattice_equation([4, 5], d_min=2, d_max=4, seed=11)
    arrays = ca.make_arrays_from_inputs(inputs, size_dict, seed=2, dtype="complex128")
    ref = np.einsum(ca.inputs_output_to_eq(inputs, output), *arrays, optimize=True)
    tree = pathfind.greedy_tree(inputs, output, size_dict, temperature=2.0, seed=9)
    m = pathfind.MI355X_C64
    t2 = pathfind.subtree_reconfigure(tree, subtree_size=9, minimize="time")
    assert modelled_seconds(t2, m) <= modelled_seconds(tree, m) * (1 + 1e-12)
    assert np.allclose(orc.contract(t2, arrays), ref, rtol=1e-10, atol=1e-12)
    # compute-only and memory-only machines reproduce the flops / write orderings
    flops_machine = pathfind.MachineModel([1.0], 1e300)
    t3 = pathfind.subtree_reconfigure(tree, subtree_size=9, minimize=flops_machine)
    t4 = pathfind.subtree_reconfigure(tree, subtree_size=9, minimize="flops")
    # (N < 16 columns are priced at N/16 of the rate, so not exactly "flops")
    assert modelled_seconds(t3, flops_machine) <= modelled_seconds(t4, flops_machine) * (1 + 1e-12)
    mem_machine = pathfind.MachineModel([1e300], 1.0)
    t5 = pathfind.subtree_reconfigure(tree, subtree_size=9, minimize=mem_machine)
    assert modelled_seconds(t5, mem_machine) <= modelled_seconds(tree, mem_machine) * (1 + 1e-12)
    assert np.allclose(orc.contract(t5, arrays), ref, rtol=1e-10, atol=1e-12)
    sliced = tree.slice(target_size=max(tree.max_size() // 4, 2))
    t6 = pathfind.subtree_reconfigure(sliced, subtree_size=8, minimize="time")
    assert set(t6.sliced_inds) == set(sliced.sliced_inds)
    assert modelled_seconds(t6, m) <= modelled_seconds(sliced, m) * (1 + 1e-12)
    assert np.allclose(orc.contract(t6, arrays), ref, rtol=1e-10, atol=1e-12)
    with pytest.raises(ValueError):
        pathfind.subtree_reconfigure(tree, minimize=pathfind.MachineModel([0.0], 1.0))


def test_tree_methods_mirror_the_reference_names():
    """tree.subtree_reconfigure(_) / tree.slice_and_reconfigure(_) with the
    reference's keyword names (core.py:2316, 2723)."""
    inputs, output, shapes, size_dict = ca.lattice_equation([3, 5], d_min=2, d_max=3, seed=2)
    arrays = ca.make_arrays_from_inputs(inputs, size_dict, seed=4, dtype="complex128")
    ref = np.einsum(ca.inputs_output_to_eq(inputs, output), *arrays, optimize=True)
    tree = pathfind.greedy_tree(inputs, output, size_dict, temperature=2.0, seed=1)
    t2 = tree.subtree_reconfigure(subtree_size=6, minimize="combo", select="max", progbar=False)
    assert t2 is not tree and t2.contraction_cost() <= tree.contraction_cost() * 64
    target = max(tree.max_size() // 4, 2)
    t3 = tree.slice_and_reconfigure(target, step_size=1, minimize="flops", reconf_opts=dict(subtree_size=6))
    assert t3.max_size() <= target and t3.nslices > 1 and tree.nslices == 1
    assert np.allclose(orc.contract(t3, arrays), ref, rtol=1e-10, atol=1e-12)
    cp = tree.copy()
    assert cp.subtree_reconfigure_(subtree_size=6) is cp
    assert cp.slice_and_reconfigure_(target, reslice=True) is cp and cp.max_size() <= target
    assert np.allclose(orc.contract(cp, arrays), ref, rtol=1e-10, atol=1e-12)


def test_refine_and_unslice_for_the_device():
    """pathfind.refine / unslice: model-guided polishing of a sliced tree; the
    modelled time to the full result only goes down, limits are respected and
    the contraction value does not change."""
    inputs, output, shapes, size_dict = ca.lattice_equation([4, 4], d_min=2, d_max=4, seed=21)
    arrays = ca.make_arrays_from_inputs(inputs, size_dict, seed=3, dtype="complex128")
    ref = np.einsum(ca.inputs_output_to_eq(inputs, output), *arrays, optimize=True)
    tree = pathfind.greedy_tree(inputs, output, size_dict, temperature=1.5, seed=4)
    width = tree.max_size()
    over = tree.slice(target_size=max(width // 16, 2))      # sliced far more than needed
    total = lambda t: pathfind.modelled_seconds(t)[0] * t.nslices
    # everything fits: all indices come back
    back = pathfind.unslice(over, max_width=width)
    assert back.nslices < over.nslices and back.max_size() <= width and total(back) <= total(over)
    # a width limit is honoured
    lim = pathfind.unslice(over, max_width=max(width // 4, 2))
    assert lim.max_size() <= max(width // 4, 2) and total(lim) <= total(over)
    # an arena limit too (nothing may be restored under a zero budget)
    assert pathfind.unslice(over, max_arena_bytes=0).nslices == over.nslices
    seen = []
    best = pathfind.refine(over, objectives=("time", "combo-64"), subtree_sizes=(4, 8), max_width=max(width // 4, 2),
                           progress=lambda rnd, obj, sz, t, v: seen.append(v))
    assert total(best) <= total(lim) * (1 + 1e-9) and best.max_size() <= max(width // 4, 2)
    assert seen == sorted(seen, reverse=True)
    assert np.allclose(orc.contract(best, arrays), ref, rtol=1e-10, atol=1e-12)
    secs, arena = pathfind.modelled_seconds(best)
    assert secs > 0 and arena > 0


def test_native_search_is_deterministic_and_valid():
    """pathfind.search: sampled greedy + reconfiguration + slicing, best draws
    refined; same arguments, same tree (also across worker processes); the value
    of the contraction is preserved and the width target met."""
    inputs, output, shapes, size_dict = ca.lattice_equation([4, 4], d_min=2, d_max=3, seed=8)
    arrays = ca.make_arrays_from_inputs(inputs, size_dict, seed=6, dtype="complex128")
    ref = np.einsum(ca.inputs_output_to_eq(inputs, output), *arrays, optimize=True)
    width = pathfind.greedy_tree(inputs, output, size_dict).max_size()
    target = max(width // 8, 2)
    log = []
    a = pathfind.search(inputs, output, size_dict, target_size=target, n_samples=6, seed=3, repeats=8,
                        refine_top=2, progress=lambda *x: log.append(x[0]))
    b = pathfind.search(inputs, output, size_dict, target_size=target, n_samples=6, seed=3, repeats=8,
                        refine_top=2, workers=2)
    assert a.get_path() == b.get_path() and list(a.sliced_inds) == list(b.sliced_inds)
    assert a.max_size() <= target and a.nslices > 1
    assert log.count("sample") == 6 and log.count("refined") == 2
    assert np.allclose(orc.contract(a, arrays), ref, rtol=1e-10, atol=1e-12)
    # more samples never hurt (the draws of the smaller search are a subset)
    c = pathfind.search(inputs, output, size_dict, target_size=target, n_samples=12, seed=3, repeats=8, refine_top=12)
    d = pathfind.search(inputs, output, size_dict, target_size=target, n_samples=6, seed=3, repeats=8, refine_top=6)
    total = lambda t: pathfind.modelled_seconds(t)[0] * t.nslices
    assert total(c) <= total(d) * (1 + 1e-9)
