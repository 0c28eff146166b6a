"""CPU suite, part 6 (round 3).

* the ordered traversal reproduces the reference's sequence (core.py:1801-1832):
  ``get_path(order=f)``, ``get_ssa_path(order=f)``, ``peak_size(order=f)`` index
  for index against fixtures frozen from the real reference
  (tests/golden/gen/make_traverse.py), incl. ``order="surface_order"``;
"""
import json
import os

import numpy as np
import pytest

import cotengra_amd as ca
from oracle import contract_ref as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

with open(os.path.join(ROOT, "tests", "golden", "traverse_cases.json"), encoding="utf-8") as _f:
    TRAVERSE = json.load(_f)["cases"]

# the same score functions as tests/golden/gen/make_traverse.py
ORDERS = {
    "size": lambda t: t.get_size,
    "flops": lambda t: t.get_flops,
    "const": lambda t: (lambda node: 0),
    "neg_extent": lambda t: (lambda node: -t.get_extent(node)),
    "size_mod7": lambda t: (lambda node: t.get_size(node) % 7),
}


def traverse_tree(case):
    inputs = [tuple(t) for t in case["inputs"]]
    tree = ca.ContractionTree.from_path(inputs, tuple(case["output"]), case["size_dict"],
                                        ssa_path=case["ssa_path"])
    for ind, project in case["sliced"]:
        tree.remove_ind_(ind, project=project)
    return tree


@pytest.mark.parametrize("case", TRAVERSE, ids=[c["name"] for c in TRAVERSE])
def test_ordered_traversal_is_the_references(case):
    tree = traverse_tree(case)
    assert [list(p) for p in tree.get_path()] == case["default"]["path"]
    assert tree.peak_size() == case["default"]["peak_size"]
    for name, make in ORDERS.items():
        f = make(tree)
        want = case["orders"][name]
        assert [list(p) for p in tree.get_path(order=f)] == want["path"], name
        assert [list(p) for p in tree.get_ssa_path(order=f)] == want["ssa_path"], name
        assert tree.peak_size(order=f) == want["peak_size"], name
        # still a valid schedule: children before parents, every contraction once
        done = set(range(tree.N))
        for p, l, r in tree.traverse(order=f):
            assert l in done and r in done and p not in done
            done.add(p)
        assert len(done) == 2 * tree.N - 1
    # explicit surface order, by name (core.py:1807-1808, 3264-3283)
    tree.set_surface_order_from_path(tree.get_ssa_path(order=tree.get_flops))
    want = case["orders"]["surface_order"]
    assert [list(p) for p in tree.get_path(order="surface_order")] == want["path"]
    assert [list(p) for p in tree.get_ssa_path_surface()] == want["ssa_path"]
    assert tree.peak_size(order="surface_order") == want["peak_size"]


def test_surface_order_without_a_path_says_why():
    inputs, output, _, size_dict = ca.lattice_equation([3, 3], d_min=2, d_max=2, seed=0)
    t = ca.ContractionTree.from_path(inputs, output, size_dict, path=ca.greedy_path(inputs, output, size_dict))
    with pytest.raises(NotImplementedError, match="set_surface_order_from_path"):
        list(t.traverse(order="surface_order"))
    with pytest.raises(ValueError):
        list(t.traverse(order="bfs"))


def test_order_changes_lifetimes_not_values():
    case = next(c for c in TRAVERSE if c["name"] == "hyper24_s0")
    tree = traverse_tree(case)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=0)
    ref = orc.contract(tree, arrays)
    for name, make in ORDERS.items():
        assert np.allclose(orc.contract(tree, arrays, order=make(tree)), ref, rtol=1e-12, atol=1e-15), name


# ---------------------------------------------------------------------- #
# fused stem pairs (cotengra_amd/stem.py): planning, tables, accounting
# ---------------------------------------------------------------------- #

import golden_util as G  # noqa: E402
from cotengra_amd import plan as P, runtime  # noqa: E402
from cotengra_amd.plan import compile_tree  # noqa: E402
from oracle.plan_interp import run_plan  # noqa: E402


@pytest.fixture
def fuse_whatever_fits(monkeypatch):
    """The pairing model prices a gather by the contiguous bytes of its load instructions and
    leaves pairs alone that would gather in small pieces; the shape tests want every pair the
    kernel can take."""
    from cotengra_amd import stem
    monkeypatch.setattr(stem, "gather_rate", lambda run_bytes: 5.4e12)


@pytest.mark.parametrize("sliced", [0, 2])
@pytest.mark.parametrize("case", range(len(G.STEM_CASES)))
def test_fused_stem_plan_matches_oracle(case, sliced, fuse_whatever_fits):
    """The fused step's tables, interpreted in numpy exactly as the kernel reads
    them (oracle/plan_interp.run_stem2), give the reference contraction; the plan
    passes the C ABI's validation; work and algorithmic bytes are those of the
    unfused plan (the roofline must not shrink because a kernel got smarter)."""
    nq, gates = G.STEM_CASES[case]
    tree = G.stem_network(nq, gates, 100 * case, sliced=sliced)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=case, dtype="complex128")
    fused = compile_tree(tree, "complex64", fuse=True, fuse_min_elems=1 << 10)
    plain = compile_tree(tree, "complex64", fuse=False)
    # (a STEM2 record with the ``one`` flag -- round 4 -- is a single step on the stem kernel, not a pair)
    n_fused = sum(s.kind == P.KIND_STEM2 and not s.stem.get("one") for s in fused.steps)
    assert len(fused.steps) == len(plain.steps) - n_fused
    # (slicing two indices of the first tensor can take a gate below K = 16)
    if sliced == 0:
        assert n_fused >= 1
    assert fused.macs_per_slice == plain.macs_per_slice
    assert fused.elems_rw_per_slice == plain.elems_rw_per_slice
    assert fused.elems_moved_per_slice <= plain.elems_rw_per_slice
    assert (fused.elems_moved_per_slice < plain.elems_rw_per_slice) == (n_fused > 0)
    assert fused.arena_elems <= plain.arena_elems
    runtime.DevicePlan(fused).close()
    fused.dtype = "complex128"   # interpret in double precision
    got = run_plan(fused, arrays)
    ref = orc.contract(tree, arrays)
    assert np.allclose(got, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("seed", range(48))
def test_random_stems_fused_plan_matches_oracle(seed, fuse_whatever_fits):
    """Random stems -- 3 to 6 gates of 4 to 7 contracted and 4 to 7 new indices each on a tensor of
    15 to 18 binary indices, contracted positions and orders drawn at random, up to two indices
    sliced -- planned with every pair the kernel can take and interpreted in numpy exactly as the
    kernel reads its tables: the reference contraction, the unfused plan's work and bytes."""
    tree = G.random_stem(seed)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=seed, dtype="complex128")
    fused = compile_tree(tree, "complex64", fuse=True, fuse_min_elems=1 << 9)
    plain = compile_tree(tree, "complex64", fuse=False)
    assert fused.macs_per_slice == plain.macs_per_slice
    assert fused.elems_rw_per_slice == plain.elems_rw_per_slice
    assert fused.elems_moved_per_slice <= plain.elems_rw_per_slice
    runtime.DevicePlan(fused).close()
    fused.dtype = "complex128"
    got = run_plan(fused, arrays)
    ref = orc.contract(tree, arrays)
    assert np.allclose(got, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())


def test_fusion_is_off_where_it_does_not_apply():
    tree = G.stem_network(16, [(3, 3), (5, 5), (5, 5)], 0)
    for dtype in ("complex128", "float32", "float64"):
        assert all(s.kind != P.KIND_STEM2 for s in compile_tree(tree, dtype, fuse=True, fuse_min_elems=1 << 10).steps)
    # default threshold: a 2^16-element stem is left alone
    assert all(s.kind != P.KIND_STEM2 for s in compile_tree(tree, "complex64").steps)
    # an extent that is not a power of two anywhere on a step keeps that step out
    inputs, output, _, size_dict = ca.lattice_equation([4, 4], d_min=3, d_max=3, seed=0)
    t3 = ca.ContractionTree.from_path(inputs, output, size_dict, path=ca.greedy_path(inputs, output, size_dict))
    assert all(s.kind != P.KIND_STEM2 for s in compile_tree(t3, "complex64", fuse=True, fuse_min_elems=1).steps)


def test_pair_model_of_the_bf16_mode(monkeypatch):
    """In the bf16 x 3 arithmetic (the default since round 4; CTG_STEM_BF16X3=0 or ``bf16x3=False``
    = fp32 products) the pair model prices the pairs' matrix work at BF16X3_SPEEDUP x the fp32 rate
    (tree refinement for that mode, tests/golden/gen/refine_bf3.py); a memory-bound pair costs the
    same either way.  The environment, when set, wins over the caller's flag."""
    from cotengra_amd import stem
    monkeypatch.setenv("CTG_STEM_BF16X3", "0")
    mfma_bound = stem.pair_seconds(2**27 * 32 * 32, 2**26 * 64 * 64, 2**32, 2**32, 8)
    mem_bound = stem.pair_seconds(2**28 * 16 * 16, 2**28 * 16 * 16, 2**32, 2**32, 16)
    assert stem.pair_seconds(2**27 * 32 * 32, 2**26 * 64 * 64, 2**32, 2**32, 8, bf16x3=True) == mfma_bound
    monkeypatch.setenv("CTG_STEM_BF16X3", "1")
    assert stem.pair_seconds(2**27 * 32 * 32, 2**26 * 64 * 64, 2**32, 2**32, 8) < 0.8 * mfma_bound
    assert stem.pair_seconds(2**28 * 16 * 16, 2**28 * 16 * 16, 2**32, 2**32, 16) >= 0.9 * mem_bound
    monkeypatch.delenv("CTG_STEM_BF16X3")
    assert stem.pair_seconds(2**27 * 32 * 32, 2**26 * 64 * 64, 2**32, 2**32, 8) < 0.8 * mfma_bound           # the default
    assert stem.pair_seconds(2**27 * 32 * 32, 2**26 * 64 * 64, 2**32, 2**32, 8, bf16x3=False) == mfma_bound


def test_fused_descriptor_is_validated():
    tree = G.stem_network(16, [(3, 3), (5, 5), (5, 5)], 0)
    plan = compile_tree(tree, "complex64", fuse=True, fuse_min_elems=1 << 10)
    st = next(s for s in plan.steps if s.kind == P.KIND_STEM2)
    for name, how in (("gA_hi", lambda t: t + (1 << 40)), ("lane_a", lambda t: t + (1 << 29)), ("kj_a", lambda t: t + (1 << 40)),
                      ("mid_row", lambda t: t * 64),
                      ("out_col", lambda t: t + (1 << 40)), ("b2_off", lambda t: t - 1)):
        keep = st.stem["tabs"][name]
        st.stem["tabs"][name] = how(keep.copy())
        with pytest.raises((runtime.CtgError, ValueError)):   # CTG_E_BOUNDS / CTG_E_INVALID
            runtime.DevicePlan(plan)
        st.stem["tabs"][name] = keep
    runtime.DevicePlan(plan).close()


def test_fused_descriptor_16_byte_gathers_are_validated(fuse_whatever_fits):
    """Where the big operand's stride-1 index is a contracted one a lane gathers two adjacent k in
    one 16-byte load (descriptor word 17): slot pairs must be adjacent, every offset even."""
    nq, gates = G.STEM_CASES[1]
    plan = compile_tree(G.stem_network(nq, gates, 100), "complex64", fuse=True, fuse_min_elems=1 << 10)
    st = next(s for s in plan.steps if s.kind == P.KIND_STEM2)
    assert st.stem["vec"] == 1
    kj = st.stem["tabs"]["kj_a"]
    assert np.all(kj[1::2] == kj[0::2] + 1) and not np.any(kj[0::2] & 1)
    runtime.DevicePlan(plan).close()

    def broken(name, how):
        keep = st.stem["tabs"][name]
        st.stem["tabs"][name] = how(keep.copy())
        with pytest.raises((runtime.CtgError, ValueError)):
            runtime.DevicePlan(plan)
        st.stem["tabs"][name] = keep

    def unpair(t):
        t[1] += 2
        return t

    def odd(t):
        t[3] += 1
        return t

    broken("kj_a", unpair)
    broken("lane_a", odd)
    broken("rt_a", lambda t: t + 1)
    st.stem["vec"] = 2
    with pytest.raises((runtime.CtgError, ValueError)):
        runtime.DevicePlan(plan)
    st.stem["vec"] = 1
    runtime.DevicePlan(plan).close()


# ---------------------------------------------------------------------- #
# checkpoint signature covers the inputs; progress counter; cache bounds
# ---------------------------------------------------------------------- #

from cotengra_amd.contractor import _Progress, inputs_digest, tree_signature  # noqa: E402


def test_checkpoint_signature_covers_the_inputs():
    inputs, output, _, size_dict = ca.lattice_equation([3, 3], d_min=2, d_max=2, seed=0)
    t = ca.ContractionTree.from_path(inputs, output, size_dict, path=ca.greedy_path(inputs, output, size_dict))
    a = ca.make_arrays_from_inputs(inputs, size_dict, seed=1)
    b = [x.copy() for x in a]
    assert inputs_digest(a) == inputs_digest(b)
    assert tree_signature(t, "complex128", arrays=a) == tree_signature(t, "complex128", arrays=b)
    b[4] = b[4] * (1 + 1e-15)   # one bit somewhere
    assert tree_signature(t, "complex128", arrays=a) != tree_signature(t, "complex128", arrays=b)
    assert tree_signature(t, "complex128", arrays=a) != tree_signature(t, "complex128")
    # dtype and shape are part of it, not only the bytes
    assert inputs_digest([np.zeros(4, "float32")]) != inputs_digest([np.zeros(2, "float64")])
    assert inputs_digest([np.zeros((2, 2))]) != inputs_digest([np.zeros(4)])


def test_progress_counter():
    seen = []
    p = _Progress(lambda done, total: seen.append((done, total)), 10)
    assert p.active
    p.update(4)
    p.update(6)
    p.close()
    assert seen == [(4, 10), (10, 10)]
    assert not _Progress(False, 10).active
    bar = _Progress(True, 3)   # tqdm is installed here: a real bar
    assert bar.active
    bar.update(3)
    bar.close()
    # gather_slices counts the slices it folds
    inputs, output, _, size_dict = ca.lattice_equation([3, 3], d_min=2, d_max=2, seed=0)
    t = ca.ContractionTree.from_path(inputs, output, size_dict, path=ca.greedy_path(inputs, output, size_dict))
    t.remove_ind_(inputs[4][0])
    arrays = ca.make_arrays_from_inputs(inputs, size_dict, seed=1)
    seen.clear()
    total = t.gather_slices(
        (orc.contract_slice(t, arrays, i) for i in range(t.nslices)),
        progbar=lambda d, n: seen.append((d, n)),
    )
    assert np.allclose(total, orc.contract(t, arrays)) and seen == [(1, 2), (2, 2)]
