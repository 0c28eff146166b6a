"""CPU suite, part 3: host-side tree semantics mirrored from the reference
(SliceInfo ordering, slice keys, paths, restore, cost model, front ends)."""
import numpy as np
import pytest

import cotengra_amd as ca
from cotengra_amd.tree import SliceInfo, get_slice_strides
from oracle import contract_ref as orc


def chain_tree():
    # ab,bc,cd->ad with sizes a2 b3 c4 d5 (SURVEY.md appendix A example)
    return ca.ContractionTree.from_path(
        ["ab", "bc", "cd"], "ad", dict(a=2, b=3, c=4, d=5), path=[(0, 1), (0, 1)]
    )


def test_ir_example_from_survey():
    t = chain_tree()
    assert tuple(orc.extract_contractions(t)) == (
        (3, 0, 1, True, ((1,), (0,)), None),
        (4, 3, 2, True, ((1,), (0,)), None),
    )
    t.remove_ind_("c")
    assert tuple(orc.extract_contractions(t)) == (
        (3, 0, 1, True, ((1,), (0,)), None),
        (4, 3, 2, True, ((), ()), None),
    )
    assert t.sliced_inds["c"] == SliceInfo(True, "c", 4, None)
    assert t.sliced_inputs == frozenset({1, 2})
    assert t.get_shapes_sliced() == ((2, 3), (3,), (5,))
    assert [t.slice_key(i) for i in range(4)] == [{"c": i} for i in range(4)]


def test_outer_sliced_first_and_keys():
    t = chain_tree()
    t.remove_ind_("b")
    t.remove_ind_("a")
    assert [si.ind for si in t.sliced_inds.values()] == ["a", "b"]  # outer (output) first
    assert get_slice_strides(t.sliced_inds) == [3, 1]
    assert [t.slice_key(i) for i in range(6)] == [
        {"a": 0, "b": 0}, {"a": 0, "b": 1}, {"a": 0, "b": 2},
        {"a": 1, "b": 0}, {"a": 1, "b": 1}, {"a": 1, "b": 2}]
    assert t.nslices == 6 and t.nchunks == 2
    t.restore_ind_("a")
    assert t.nslices == 3 and t.sliced_inputs == frozenset({0, 1})
    t.unslice_all_()
    assert t.nslices == 1 and not t.sliced_inds and t.sliced_inputs == frozenset()


def test_projection_and_errors():
    t = chain_tree()
    p = t.remove_ind("b", project=2)
    assert p.nslices == 1 and p.slice_key(0) == {"b": 2}
    with pytest.raises(ValueError):
        p.remove_ind("b")
    with pytest.raises(ValueError):
        ca.ContractionTree.from_path(["ab", "bc"], "ac", dict(a=2, b=2, c=2))
    # a three-tensor step is expanded into pairwise merges
    t3 = ca.ContractionTree.from_path(["ab", "bc", "cd"], "ad", dict(a=2, b=2, c=2, d=2), path=[(0, 1, 2)])
    assert t3.is_complete() and len(list(t3.traverse())) == 2


def test_cost_model_and_paths():
    inputs, output, shapes, size_dict = ca.lattice_equation([4, 4], d_min=3)
    path = ca.greedy_path(inputs, output, size_dict)
    t = ca.ContractionTree.from_path(inputs, output, size_dict, path=path)
    assert t.is_complete()
    # get_path() follows the depth-first traversal, so it is a fixed point
    assert ca.ContractionTree.from_path(inputs, output, size_dict, path=t.get_path()).get_path() == t.get_path()
    assert t.contraction_cost() == ca.ContractionTree.from_path(
        inputs, output, size_dict, path=t.get_path()).contraction_cost()
    t2 = ca.ContractionTree.from_path(inputs, output, size_dict, ssa_path=t.get_ssa_path())
    assert t2.get_path() == t.get_path()
    assert t.total_flops("complex64") == 4 * t.contraction_cost()
    assert t.total_flops("float32") == 2 * t.contraction_cost()
    assert t.peak_size() >= t.max_size()
    # children before parents, left subtree spans at least as many leaves
    seen = set(range(t.N))
    for p, l, r in t.traverse():
        assert l in seen and r in seen
        assert t.get_extent(l) >= t.get_extent(r)
        seen.add(p)
    ordered = list(t.traverse(order=lambda n: t.get_size(n)))
    assert sorted(p for p, _, _ in ordered) == sorted(p for p, _, _ in t.traverse())


def test_slice_arrays_views_and_sum():
    inputs, output, shapes, size_dict = ca.lattice_equation([3, 3], d_min=2)
    t = ca.ContractionTree.from_path(inputs, output, size_dict, path=ca.greedy_path(inputs, output, size_dict))
    arrays = ca.make_arrays_from_inputs(inputs, size_dict, seed=1)
    full = orc.contract(t, arrays)
    t.remove_ind_(inputs[4][1])
    sl = t.slice_arrays(arrays, 1)
    assert all(s.base is not None or s is a for s, a in zip(sl, arrays))  # views, not copies
    assert np.allclose(sum(orc.contract_slice(t, arrays, i) for i in range(t.nslices)), full)


def test_lattice_and_arrays_match_reference_semantics():
    inputs, output, shapes, size_dict = ca.lattice_equation([2, 3])
    assert [''.join(t) for t in inputs] == ['ab', 'cbd', 'ed', 'af', 'cfg', 'eg']
    xs = ca.make_arrays_from_inputs(inputs, size_dict, seed=42, dtype="complex64")
    assert all(abs(np.linalg.norm(x) - 1) < 1e-6 for x in xs)
    rng = np.random.default_rng(42)
    first = rng.normal(size=shapes[0]) + 1j * rng.normal(size=shapes[0])
    assert np.allclose(xs[0], (first / np.linalg.norm(first)).astype("complex64"))


def test_greedy_path_handles_hyper_and_disconnected():
    for eq in ("ab,bc,cd->ad", "ab,ab,ab->ab", "a,b,c->abc", "abx,bcx,cdx->adx", "ab,cd->", "aab,bc->ac"):
        inputs, output = ca.eq_to_inputs_output(eq)
        sd = {ix: 2 + (ord(ix) % 3) for t in inputs for ix in t}
        t = ca.array_contract_tree(inputs, output, sd)
        xs = [np.random.default_rng(0).normal(size=[sd[i] for i in term]) for term in inputs]
        assert np.allclose(orc.contract(t, xs), np.einsum(eq, *xs))


def test_api_compat_helpers(capsys):
    from cotengra_amd.interface import Variadic, Via

    t = chain_tree()
    t.sort_contraction_indices("flops")   # (parity with the reference: tests/test_host_round4.py)
    t.print_contractions()
    out = capsys.readouterr().out
    assert out.count("cost:") == 2 and "ab,bc->ac" in out
    v = Variadic(lambda arrays, scale=1: scale * sum(arrays), scale=2)
    assert v(1, 2, 3) == 12
    w = Via(lambda *xs: sum(xs), lambda x: x + 1, lambda y: -y)
    assert w(1, 2) == -5


def test_from_path_expands_multi_tensor_steps():
    """A path step naming three or more tensors (opt_einsum allows it) becomes
    pairwise merges; the contraction value is unchanged."""
    import numpy as np
    from oracle import contract_ref as orc

    inputs = [("a", "b"), ("b", "c"), ("c", "d"), ("d", "e"), ("e", "a")]
    size_dict = {"a": 3, "b": 4, "c": 2, "d": 5, "e": 3}
    tree = ca.ContractionTree.from_path(inputs, (), size_dict, path=[(0, 1, 2), (0, 1, 2)])
    assert tree.is_complete() and len(list(tree.traverse())) == 4
    arrays = ca.make_arrays_from_inputs(inputs, size_dict, seed=3, dtype="float64")
    ref = np.einsum("ab,bc,cd,de,ea->", *arrays)
    assert abs(orc.contract(tree, arrays) - ref) < 1e-12 * abs(ref)
    # ssa form as well
    tree2 = ca.ContractionTree.from_path(inputs, (), size_dict, ssa_path=[(0, 1, 2, 3), (4, 5)])
    assert abs(orc.contract(tree2, arrays) - ref) < 1e-12 * abs(ref)


def test_compile_refuses_unsliceably_wide_trees():
    """A tree whose single-slice intermediates cannot fit any device fails with a
    clear MemoryError at plan time instead of exhausting the host."""
    import cotengra_amd as ca
    from cotengra_amd.plan import compile_tree

    inputs = [tuple(f"i{j}" for j in range(20)), tuple(f"k{j}" for j in range(20))]
    sd = {ix: 2 for t in inputs for ix in t}
    tree = ca.ContractionTree.from_path(inputs, inputs[0] + inputs[1], sd, path=[(0, 1)])
    with pytest.raises(MemoryError, match="slice the tree"):
        compile_tree(tree, "complex64")
