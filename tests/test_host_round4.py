"""CPU suite, part 7 (round 4).

* ``ContractionTree.sort_contraction_indices`` reproduces the reference's index orders
  (core.py:3421-3506): ``get_inds`` of every node and the IR derived from them against
  fixtures frozen from the real reference (tests/golden/gen/make_golden_r4.py), for four
  priorities x the two switches, and a two-call sequence with ``reset=False``;
* the per-op plug-in after sorting hands the caller the sorted tensordot axes / perms and
  still gets the reference-frozen value (numpy as the caller's backend);
* the one-shot expression cache never holds its own lock and a contractor's at once
  (lock-order inversion);
* boolean ``CTG_*`` switches: "0" and "" mean off.
"""
import hashlib
import json
import os
import threading

import numpy as np
import pytest

import cotengra_amd as ca
from oracle import contract_ref as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

with open(os.path.join(ROOT, "tests", "golden", "sorted_inds_cases.json"), encoding="utf-8") as _f:
    SORTED = json.load(_f)["cases"]
EXPECTED = np.load(os.path.join(ROOT, "tests", "golden", "sorted_inds_expected.npz"))


def ir_hash(ops):
    return hashlib.sha256("\n".join(sorted(repr(tuple(op)) for op in ops)).encode()).hexdigest()


def build(case):
    inputs = [tuple(t) for t in case["inputs"]]
    tree = ca.ContractionTree.from_path(inputs, tuple(case["output"]), case["size_dict"], ssa_path=case["ssa_path"])
    for ind, project in case["sliced"]:
        tree.remove_ind_(ind, project=project)
    return tree


def node_inds(tree):
    return [[list(tree.get_inds_tuple(n)) for n in plr] for plr in tree.traverse()]


@pytest.mark.parametrize("case", SORTED, ids=[c["name"] for c in SORTED])
def test_sorted_contraction_indices_are_the_references(case):
    tree = build(case)
    assert node_inds(tree) == case["default"]["inds"]
    assert ir_hash(orc.extract_contractions(tree)) == case["default"]["ir"]
    n_changed = 0
    for want in case["sorted"]:
        t = build(case)
        t.get_tensordot_axes(t.root)   # something cached beforehand must not survive the reset
        t.sort_contraction_indices(priority=want["priority"], make_output_contig=want["make_output_contig"],
                                   make_contracted_contig=want["make_contracted_contig"])
        key = (want["priority"], want["make_output_contig"], want["make_contracted_contig"])
        assert node_inds(t) == want["inds"], key
        assert ir_hash(orc.extract_contractions(t)) == want["ir"], key
        n_changed += want["inds"] != case["default"]["inds"]
        # legs are untouched: every node still carries exactly its legs
        for p, l, r in t.traverse():
            for n in (p, l, r):
                assert sorted(t.get_inds_tuple(n)) == sorted(t.get_legs(n))
        # the root keeps the caller's output order
        assert t.get_inds_tuple(t.root) == tuple(ix for ix in t.output if ix not in t.sliced_inds)
    assert n_changed >= 3   # the fixture exercises the method
    # a second call without reset starts from the first one's orders (core.py:3455-3456, 3503-3505)
    t = build(case)
    t.sort_contraction_indices(priority="size")
    t.contraction_cores["sentinel"] = object()
    t.sort_contraction_indices(priority="leaves", make_output_contig=False, reset=False)
    assert node_inds(t) == case["sequence"]["inds"]
    assert not t.contraction_cores
    with pytest.raises(ValueError):
        t.sort_contraction_indices(priority="bogus")


@pytest.mark.parametrize("name", sorted(EXPECTED.files))
def test_sorted_tree_through_the_per_op_plugin(name):
    """``implementation=(einsum, tensordot)`` walks the SORTED index algebra (axes, perms,
    equations differ from the default order's) and lands on the value the reference got
    from its sorted tree."""
    case = next(c for c in SORTED if c["name"] == name)
    tree = build(case)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=case["seed"], dtype="complex128")
    calls = {"tensordot": [], "einsum": []}

    def np_einsum(eq, *xs):
        calls["einsum"].append(eq)
        return np.einsum(eq, *xs)

    def np_tensordot(a, b, axes):
        calls["tensordot"].append(tuple(map(tuple, axes)))
        return np.tensordot(a, b, axes)

    plain = tree.contract(arrays, implementation=(np_einsum, np_tensordot))
    seen_plain = {k: list(v) for k, v in calls.items()}
    axes_before = [tree.get_tensordot_axes(p) if tree.get_can_dot(p) else tree.get_einsum_eq(p)
                   for p, _, _ in tree.traverse()]
    calls["tensordot"].clear()
    calls["einsum"].clear()
    tree.sort_contraction_indices()
    got = tree.contract(arrays, implementation=(np_einsum, np_tensordot))
    want = EXPECTED[name]
    assert np.asarray(got).shape == want.shape
    assert np.allclose(got, want, rtol=1e-10, atol=1e-13 * max(1.0, float(np.abs(want).max())))
    assert np.allclose(plain, want, rtol=1e-10, atol=1e-13 * max(1.0, float(np.abs(want).max())))
    axes_after = [tree.get_tensordot_axes(p) if tree.get_can_dot(p) else tree.get_einsum_eq(p)
                  for p, _, _ in tree.traverse()]
    # the plug-in saw the sorted index algebra (where sorting moved a contracted axis at all)
    assert (calls != seen_plain) == (axes_after != axes_before)
    # and the oracle run on the sorted tree agrees (its IR is the reference's, checked above)
    assert np.allclose(orc.contract(tree, arrays), want, rtol=1e-10, atol=1e-13 * max(1.0, float(np.abs(want).max())))


def test_descend_visits_parents_first():
    case = next(c for c in SORTED if c["name"] == "randreg30_s0")
    tree = build(case)
    for mode in ("dfs", "bfs"):
        seen = {tree.root}
        n = 0
        for p, l, r in tree.descend(mode):
            assert p in seen and (l, r) == tuple(tree.children[p])
            seen.update((l, r))
            n += 1
        assert n == tree.N - 1
    with pytest.raises(ValueError):
        list(tree.descend("sideways"))


def test_expression_front_end_sorts_when_asked():
    inputs, output, shapes, size_dict = ca.lattice_equation([3, 3], d_min=2, d_max=3, seed=1)
    from cotengra_amd.interface import array_contract_tree

    plain = array_contract_tree(inputs, output, size_dict, "greedy")
    srt = array_contract_tree(inputs, output, size_dict, "greedy", sort_contraction_indices=True)
    plain.sort_contraction_indices()
    assert node_inds(srt) == node_inds(plain)


# ---- the expression cache and its locks ----------------------------------------------------

class _FakeExec:
    def __init__(self, n):
        self.n, self.closed = n, False

    def device_bytes(self):
        return self.n

    def close(self):
        self.closed = True


class _FakeFn:
    def __init__(self, nbytes):
        self._lock = threading.RLock()
        self._execs = {("complex64", 0, False): {"exec": _FakeExec(nbytes)}}
        self._plans = {}
        self.closed = False

    def close(self):
        with self._lock:
            for st in self._execs.values():
                st["exec"].close()
            self._execs.clear()
            self.closed = True


def _fake_expr(nbytes):
    from cotengra_amd import interface

    e = interface.ContractExpression.__new__(interface.ContractExpression)
    e.tree, e.fn, e._cached, e._bytes = None, _FakeFn(nbytes), True, nbytes
    return e


def test_cache_never_waits_for_a_contractor_under_its_own_lock(monkeypatch):
    """Thread A is inside expression X (holds X's contractor lock) and runs out of memory:
    ``evict_expression_cache``.  Thread B trims the cache at the same time and picks X as its
    victim.  Before round 4 B closed X under the cache lock (waiting for A) while A waited
    for the cache lock: a deadlock.  Now B only unlinks X under the cache lock and closes it
    afterwards; A never blocks on a contractor lock."""
    from cotengra_amd import interface

    monkeypatch.setattr(interface, "_EXPR_CACHE", type(interface._EXPR_CACHE)())
    monkeypatch.setattr(interface, "_EXPR_CACHE_BYTES", 100)
    x, y, z = _fake_expr(80), _fake_expr(80), _fake_expr(80)
    interface._EXPR_CACHE.update({"x": x, "y": y})
    a_inside, b_unlinked, done = threading.Event(), threading.Event(), []

    def thread_a():
        with x.fn._lock:                      # inside X: upload -> run -> fetch
            a_inside.set()
            assert b_unlinked.wait(10)        # B has taken X out of the cache and wants to close it
            interface.evict_expression_cache(keep=x.fn)   # must not deadlock
            done.append("a")

    def thread_b():
        assert a_inside.wait(10)
        with interface._EXPR_LOCK:
            interface._EXPR_CACHE["z"] = z
            victims = interface._trim_expression_cache(keep=z)
        assert x in victims                   # X, least recently used, is evicted ...
        b_unlinked.set()
        interface._close_all(victims)         # ... and closed once A is out of it
        done.append("b")

    ta, tb = threading.Thread(target=thread_a), threading.Thread(target=thread_b)
    ta.start()
    tb.start()
    ta.join(20)
    tb.join(20)
    assert not ta.is_alive() and not tb.is_alive(), "deadlock"
    assert sorted(done) == ["a", "b"]
    assert x.fn.closed and list(interface._EXPR_CACHE.values()) == [z] or y.fn.closed


def test_evict_skips_expressions_in_use(monkeypatch):
    from cotengra_amd import interface

    monkeypatch.setattr(interface, "_EXPR_CACHE", type(interface._EXPR_CACHE)())
    busy, idle, keep = _fake_expr(10), _fake_expr(10), _fake_expr(10)
    interface._EXPR_CACHE.update({"busy": busy, "idle": idle, "keep": keep})
    got = []

    def other():
        got.append(interface.evict_expression_cache(keep=keep.fn))

    with busy.fn._lock:                        # another thread is inside `busy`
        th = threading.Thread(target=other)
        th.start()
        th.join(10)
        assert not th.is_alive(), "evict blocked on a contractor lock"
    assert got == [True]
    assert list(interface._EXPR_CACHE.values()) == [keep]
    assert not idle.fn._execs and busy.fn._execs   # the idle one freed, the busy one left alone


def test_device_bytes_tolerates_a_busy_contractor():
    e = _fake_expr(123)
    assert e.device_bytes() == 123
    e._bytes = 7
    held = threading.Event()
    release = threading.Event()

    def hold():
        with e.fn._lock:
            held.set()
            release.wait(10)

    th = threading.Thread(target=hold)
    th.start()
    assert held.wait(10)
    assert e.device_bytes() == 7               # last known value, no waiting
    release.set()
    th.join(10)


@pytest.mark.parametrize("value,on", [(None, True), ("", False), ("0", False), ("1", True), ("yes", True)])
def test_boolean_switches_treat_zero_as_off(monkeypatch, value, on):
    """CTG_STEM_BF16X3: unset = the default arithmetic of the fused pairs (bf16 x 3 since round 4)
    unless the caller's flag says otherwise; set = it decides, "0" and "" meaning off -- the rule
    the C side's launcher applies at every launch (csrc/ctg_stem.hip: stem2_bf3)."""
    from cotengra_amd import stem

    if value is None:
        monkeypatch.delenv("CTG_STEM_BF16X3", raising=False)
        assert stem.bf16x3_mode(False) is False and stem.bf16x3_mode(True) is True
    else:
        monkeypatch.setenv("CTG_STEM_BF16X3", value)
        assert stem.bf16x3_mode(not on) is on      # the environment wins
    assert stem.bf16x3_mode() is on


# ---- single stem steps (round 4): the stem kernel's first half alone ------------------------

import golden_util as _G  # noqa: E402

ONE_CASES = _G.ONE_CASES


@pytest.fixture
def take_every_single(monkeypatch):
    """(every single step the kernel can take, whatever the model thinks it gains)"""
    from cotengra_amd import stem

    monkeypatch.setattr(stem, "gather_rate", lambda run_bytes: 5.4e12)
    monkeypatch.setattr(stem, "MIN_GAIN", -1e9)


@pytest.mark.parametrize("case", range(len(ONE_CASES)))
@pytest.mark.parametrize("sliced", [0, 2])
def test_single_stem_steps_plan_semantics(case, sliced, take_every_single):
    """A large step no pair took is planned as a STEM2 record with the ``one`` flag
    (stem.build_stem_one); the numpy interpreter of the plan (oracle/plan_interp.py) executes it from
    the very tables the kernel reads and lands on the oracle's result; the C ABI accepts the
    descriptor and refuses corrupted ones."""
    import golden_util as G
    from cotengra_amd import plan as P, runtime
    from oracle import plan_interp

    nq, gates = ONE_CASES[case]
    tree = G.stem_network(nq, gates, 300 + case, sliced=sliced)
    plan = P.compile_tree(tree, "complex64", fuse=True, fuse_min_elems=1 << 10)
    ones = [s for s in plan.steps if s.kind == P.KIND_STEM2 and s.stem.get("one")]
    if sliced and not ones:
        pytest.skip("the sliced indices are contracted ones of the step: no longer a shape the kernel takes")
    assert len(ones) == 1, [s.label for s in plan.steps]
    s1 = ones[0]
    assert s1.b2 is None and s1.stem["K2"] == 0 and s1.elems_moved == s1.elems_rw
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=case, dtype="complex128")
    ref = orc.contract(tree, arrays)
    plan128 = P.compile_tree(tree, "complex64", fuse=True, fuse_min_elems=1 << 10)
    plan128.dtype = "complex128"   # (the interpreter only needs the tables)
    got = plan_interp.run_plan(plan128, arrays)
    assert np.allclose(got, ref, rtol=1e-10, atol=1e-13 * np.abs(ref).max())
    # the library validates the descriptor (host-only) ...
    runtime.DevicePlan(plan).close()
    # ... and refuses corrupted ones: the flag out of range, a second step smuggled in, tables
    # that reach outside the result / the operand
    st = s1.stem
    for key, value in (("one", 2), ("K2", 32), ("rows2", 32)):
        keep = st[key]
        st[key] = value
        with pytest.raises((runtime.CtgError, ValueError)):
            runtime.DevicePlan(plan)
        st[key] = keep
    for name, how in (("out_row", lambda t: t + (1 << 40)), ("out_col", lambda t: t + (1 << 40)),
                      ("gA_hi", lambda t: t + (1 << 40)), ("lane_a", lambda t: t + (1 << 29))):
        keep = st["tabs"][name]
        st["tabs"][name] = how(keep.copy())
        with pytest.raises((runtime.CtgError, ValueError)):
            runtime.DevicePlan(plan)
        st["tabs"][name] = keep
    runtime.DevicePlan(plan).close()


def test_singles_are_chosen_by_the_model(monkeypatch):
    """On the m20 headline tree the large steps the pairing leaves alone (chains of odd length) go to
    the stem kernel's first half in the bf16 x 3 arithmetic -- there a lone K = 32 ... 128 step is
    memory-bound and the tiled fp32 kernel is not -- and stay on the tiled kernel in fp32 arithmetic,
    where the model sees nothing to gain; CTG_NO_STEM_ONE=1 switches them off."""
    import os

    from cotengra_amd import plan as P

    tree = ca.tree_from_record(ca.load_network(os.path.join(ROOT, "tests", "golden", "trees", "sycamore_m20_native.json")))
    monkeypatch.delenv("CTG_STEM_BF16X3", raising=False)
    plan = P.compile_tree(tree, "complex64")
    ones = [s for s in plan.steps if s.kind == P.KIND_STEM2 and s.stem.get("one")]
    assert len(ones) >= 3 and any(s.a.size >= 2**30 for s in ones)
    # (a single step needs >= 32 output columns: a lone 16 x 16 step stays on the streaming kernel, memory-bound there too)
    left = [s for s in plan.steps if s.kind == P.KIND_PAIR and not s.invariant and s.a.size >= 2**28 and s.K >= 16 and s.N >= 32]
    assert not left, [s.label for s in left]
    # a pair whose tile has no room for the bf16 limb planes would multiply in fp32: priced so, and not chosen here
    pairs = [s for s in plan.steps if s.kind == P.KIND_STEM2 and not s.stem.get("one")]
    assert pairs and all(s.stem["bf3_fits"] for s in pairs)
    fp32 = P.compile_tree(tree, "complex64", stem_bf16x3=False)
    assert not [s for s in fp32.steps if s.kind == P.KIND_STEM2 and s.stem.get("one")]
    monkeypatch.setenv("CTG_NO_STEM_ONE", "1")
    off = P.compile_tree(tree, "complex64")
    assert not [s for s in off.steps if s.kind == P.KIND_STEM2 and s.stem.get("one")]
    assert plan.macs_per_slice == fp32.macs_per_slice == off.macs_per_slice


def test_bf16x3_tile_fit_is_part_of_the_pair_model():
    """``stem.bf16x3_fits`` restates the C side's LDS sizing of the limb planes (csrc/ctg_stem.hip
    stem2_lds_bytes_bf3, the condition of stem2_bf3): 32 32 | 64 128 on 128 rows needs 185 KB and
    multiplies in fp32 whatever the mode; the model prices it at the fp32 rate."""
    from cotengra_amd import stem

    assert stem.bf16x3_fits(32, 32, 64, 64, 128) and stem.bf16x3_fits(16, 16, 128, 64, 32)
    assert not stem.bf16x3_fits(32, 32, 64, 128, 128) and not stem.bf16x3_fits(16, 16, 128, 64, 64)
    args = (2**25 * 32 * 32, 2**24 * 64 * 128, 2**30, 2**31, 16)
    fits, not_fits = stem.pair_seconds(*args, bf16x3=True), stem.pair_seconds(*args, bf16x3=True, bf3_fits=False)
    assert not_fits == stem.pair_seconds(*args, bf16x3=False) > fits


# ---- three-step tiles (opt-in: CTG_STEM_TRIPLES; stem.geometry3 / build_stem_triple) -------------------

TRIPLE_SEEDS = (0, 27, 34, 36, 41, 42, 43, 45)   # random stems (golden_util.random_stem) with a chain of three that fits


@pytest.fixture
def take_every_triple(monkeypatch):
    """(every three-step tile that fits, whatever the model thinks it gains -- measured: nothing)"""
    from cotengra_amd import stem

    monkeypatch.setattr(stem, "gather_rate", lambda run_bytes: 5.4e12)
    monkeypatch.setattr(stem, "TRIPLE_STAGE_RATE", {n: 1e15 for n in stem.TRIPLE_STAGE_RATE})


def test_three_step_tiles_are_opt_in(monkeypatch, take_every_triple):
    """Without CTG_STEM_TRIPLES no plan holds a middle stage, however cheap the model finds one."""
    import golden_util as G
    from cotengra_amd import plan as P

    monkeypatch.delenv("CTG_STEM_TRIPLES", raising=False)
    for seed in TRIPLE_SEEDS[:4]:
        plan = P.compile_tree(G.random_stem(seed), "complex64", fuse=True, fuse_min_elems=1 << 9)
        assert not [s for s in plan.steps if s.kind == P.KIND_STEM2 and s.stem.get("KM")]


@pytest.mark.parametrize("seed", TRIPLE_SEEDS)
def test_three_step_tile_plan_semantics(seed, monkeypatch, take_every_triple):
    """Three consecutive stem steps as ONE record with a middle stage: the numpy interpreter of the
    plan executes it from the very tables the kernel reads (first intermediate, middle product, second
    intermediate written through mid2_row / mid2_col, last product) and lands on the oracle's result;
    work and algorithmic bytes are those of the unfused plan, the moved bytes fewer."""
    import golden_util as G
    from cotengra_amd import plan as P
    from oracle import plan_interp

    monkeypatch.setenv("CTG_STEM_TRIPLES", "any")   # (every tile that fits; the library's kernel list is not asked)
    tree = G.random_stem(seed)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=seed, dtype="complex128")
    fused = P.compile_tree(tree, "complex64", fuse=True, fuse_min_elems=1 << 9)
    plain = P.compile_tree(tree, "complex64", fuse=False)
    tri = [s for s in fused.steps if s.kind == P.KIND_STEM2 and s.stem.get("KM")]
    assert len(tri) == 1, [s.label for s in fused.steps]
    st = tri[0].stem
    assert (1 << st["nr1"]) * st["N1"] == st["rowsM"] * st["KM"] and st["rowsM"] * st["NM"] == st["rows2"] * st["K2"]
    assert st["lds_bytes"] <= 160 * 1024 and st["itemsM"] % 8 == 0 and st["items"] % 8 == 0
    assert fused.macs_per_slice == plain.macs_per_slice and sum(st["macs3"]) == tri[0].macs
    assert fused.elems_rw_per_slice == plain.elems_rw_per_slice
    assert fused.elems_moved_per_slice < plain.elems_rw_per_slice
    fused.dtype = "complex128"
    got = plan_interp.run_plan(fused, arrays)
    ref = orc.contract(tree, arrays)
    assert np.allclose(got, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("seed", (0, 34, 41, 45))
def test_three_step_tile_records_validate(seed, monkeypatch, take_every_triple):
    """CTG_STEM_TRIPLES=1: the planner asks the library which shapes it has kernels for
    (ctg_stem_triple_instantiated, ABI 5) and the C ABI accepts exactly such records -- and refuses a
    middle stage whose shape, tables or operand do not fit."""
    import golden_util as G
    from cotengra_amd import plan as P, runtime

    monkeypatch.setenv("CTG_STEM_TRIPLES", "1")
    if runtime.load().ctg_stem_triple_instantiated(1, 0, 1, 2, 1, 1, 1, 1, 0) != 1:
        # (round 5: the product library is built without the three-step kernels -- slower than pairs on every
        # tree; tools/build_variants.py triples=-DCTG_STEM_TRIPLES_BUILD has them) the planner then emits none
        plan = P.compile_tree(G.random_stem(seed), "complex64", fuse=True, fuse_min_elems=1 << 9)
        assert not [s for s in plan.steps if s.kind == P.KIND_STEM2 and s.stem.get("KM")]
        pytest.skip("library built without three-step tiles (CTG_STEM_TRIPLES_BUILD)")
    plan = P.compile_tree(G.random_stem(seed), "complex64", fuse=True, fuse_min_elems=1 << 9)
    tri = [s for s in plan.steps if s.kind == P.KIND_STEM2 and s.stem.get("KM")]
    assert len(tri) == 1
    runtime.DevicePlan(plan).close()
    st = tri[0].stem
    for key, value in (("KM", 48), ("NM", 8), ("rowsM", st["rowsM"] * 2), ("ldM", st["KM"]), ("ngM", 3)):
        keep = st[key]
        st[key] = value
        with pytest.raises((runtime.CtgError, ValueError)):
            runtime.DevicePlan(plan)
        st[key] = keep
    for name, how in (("mid2_row", lambda t: t + (1 << 20)), ("mid2_col", lambda t: t + (1 << 20)),
                      ("mid_row", lambda t: t + (1 << 20)), ("bm_off", lambda t: t + (1 << 40))):
        keep = st["tabs"][name]
        st["tabs"][name] = how(keep.copy())
        with pytest.raises((runtime.CtgError, ValueError)):
            runtime.DevicePlan(plan)
        st["tabs"][name] = keep
    runtime.DevicePlan(plan).close()
    lib = runtime.load()
    from cotengra_amd import stem

    class Geo:   # (the fields triple_shape reads)
        pass

    assert lib.ctg_stem_triple_instantiated(1, 0, 1, 2, 1, 1, 1, 1, 0) == 1       # seed 0's
    assert lib.ctg_stem_triple_instantiated(1, 1, 1, 7, 3, 5, 2, 2, 0) == 0
    assert stem.triples_enabled()


# ---- slice groups (plan.choose_slice_group; ctg_plan_desc.slice_group, ctg_exec_run_slice_list) ---------

@pytest.fixture
def groups_on_small_trees(monkeypatch):
    """(the planner keeps slice groups to trees whose slices are long launch sequences; the narrowed
    fixtures the oracle can run are not)"""
    from cotengra_amd import plan as P

    monkeypatch.delenv("CTG_SLICE_GROUPS", raising=False)
    monkeypatch.setattr(P, "GROUP_MIN_WIDTH", 1)
    monkeypatch.setattr(P, "GROUP_MIN_SAVING", 0.0)


@pytest.mark.parametrize("fixture", ["sycamore_m20_w32_r4.json", "sycamore_m20_native.json", "sycamore_m20_w33_bf3.json"])
def test_slice_groups_plan_semantics(fixture, groups_on_small_trees):
    """Slices that differ only in the plan's group indices share every step that depends on none of
    them.  The numpy interpreter of the plan visits the slices group by group, skips the shared steps
    for all but the first slice of a group -- reading what they left in the arena, at the offsets the
    kernels would -- and lands on the sum of the oracle's slices: whole groups, partial groups and lone
    slices in one list.  The C ABI validates the plan, and rejects one whose kept tensors are not safe."""
    import copy

    from cotengra_amd import plan as P, runtime
    from oracle import plan_interp
    from test_tree_fixtures import narrowed

    tree = ca.tree_from_record(ca.load_network(os.path.join(ROOT, "tests", "golden", "trees", fixture)))
    small = narrowed(tree, 10)
    arrays = ca.make_arrays_from_inputs(small.inputs, small.size_dict, seed=42, dtype="complex128", rescale=True)
    plan = P.compile_tree(small, "complex64", fuse_min_elems=1 << 6)
    shared = [s for s in plan.steps if s.group]
    assert plan.group_inds and shared and plan.group_size == 2 ** len(plan.group_inds)
    assert not any(s.group and s.invariant for s in plan.steps)
    # the helpers of the plan and of the interpreter agree on who shares with whom
    for sid in (0, 37, 12345, 999999):
        members = plan_interp.group_members(plan, sid)
        assert members == plan.group_ids(plan.group_of(sid)) and sid in members
        assert len({plan_interp.group_key(plan, i) for i in members}) == 1
    ids = (plan_interp.group_members(plan, 37) + plan_interp.group_members(plan, 12345)[:5]
           + plan_interp.group_members(plan, 999999) + [5, 900])
    assert len({plan.group_of(i) for i in ids}) < len(ids)
    runtime.DevicePlan(plan).close()
    plan128 = copy.copy(plan)
    plan128.dtype = "complex128"
    got = complex(np.asarray(plan_interp.run_plan(plan128, arrays, slice_ids=ids[::-1])))
    ref = sum(complex(orc.contract_slice(small, arrays, i)) for i in ids)
    assert abs(got - ref) <= 1e-10 * abs(ref)
    # without groups: the same plan but for the sharing classes and where the kept tensors live
    os.environ["CTG_SLICE_GROUPS"] = "0"
    try:
        flat = P.compile_tree(small, "complex64", fuse_min_elems=1 << 6)
    finally:
        del os.environ["CTG_SLICE_GROUPS"]
    assert flat.group_size == 1 and not [s for s in flat.steps if s.group]
    assert flat.macs_per_slice == plan.macs_per_slice and len(flat.steps) == len(plan.steps)
    # corrupted: group flags gone while steps still claim to be shared; a kept tensor moved into the
    # recycled part of the arena (another step writes there)
    keep = plan.slice_group
    plan.slice_group = [0] * len(keep)
    with pytest.raises((runtime.CtgError, ValueError)):
        runtime.DevicePlan(plan)
    plan.slice_group = keep
    # a kept tensor moved onto the output range of a per-slice step that nobody else writes between the shared
    # step and the kept tensor's first per-slice reader (the value would be alive there): refused
    def first_writer(g, skip, lo, hi):
        return next((i for i, t in enumerate(plan.steps) if i > g and t.kind != P.KIND_ACCUM and t.c.space == P.SPACE_ARENA
                     and t.c is not skip and t.c.offset < hi and lo < t.c.offset + t.c.size), len(plan.steps))

    kept = victim = None
    for cand in shared:
        g = plan.steps.index(cand)
        readers = [i for i, t in enumerate(plan.steps) if i > g and not t.group and not t.invariant and
                   t.kind != P.KIND_ACCUM and any(op is cand.c for op in (t.a, t.b, getattr(t, "b2", None)))]
        if not readers or cand.c.space != P.SPACE_ARENA:
            continue
        victim = next((t for t in plan.steps if not t.group and not t.invariant and t.kind in (P.KIND_PAIR, P.KIND_STEM2)
                       and t.c.space == P.SPACE_ARENA and t.c.size >= cand.c.size and t.c is not cand.c
                       and first_writer(g, cand.c, t.c.offset, t.c.offset + cand.c.size) > readers[0]), None)
        if victim is not None:
            kept = cand
            break
    assert kept is not None
    old = kept.c.offset
    kept.c.offset = victim.c.offset
    with pytest.raises((runtime.CtgError, ValueError)):
        runtime.DevicePlan(plan)
    kept.c.offset = old
    runtime.DevicePlan(plan).close()


def test_slice_groups_thresholds(monkeypatch):
    """By default the planner groups slices on trees whose slices are sequences of large launches (width
    >= 2^28) as soon as 2 % of a slice is shared; on small trees -- whose slices go out many per launch,
    whole groups per launch then -- only when at least 15 % is, and never next to fused stem steps."""
    import golden_util as G
    from cotengra_amd import plan as P
    from test_tree_fixtures import narrowed

    monkeypatch.delenv("CTG_SLICE_GROUPS", raising=False)
    tree = ca.tree_from_record(ca.load_network(os.path.join(ROOT, "tests", "golden", "trees", "sycamore_m20_w32_r4.json")))
    small = narrowed(tree, 10)
    monkeypatch.setattr(P, "GROUP_MIN_SAVING_SMALL", 0.99)
    assert P.compile_tree(small, "complex64").group_size == 1
    monkeypatch.setattr(P, "GROUP_MIN_SAVING_SMALL", 0.15)
    c5 = G.tree_of(next(c for c in G.cases("tree") if c["name"] == "C5_hyper200"))
    p5 = P.compile_tree(c5, "complex64")
    assert p5.group_size == 3 and not any(s.kind == P.KIND_STEM2 for s in p5.steps)
    wide = P.compile_tree(tree, "complex64")
    assert wide.group_size == 4 and 0.05 < wide.macs_shared_per_group / wide.macs_per_slice < 0.5
    monkeypatch.setenv("CTG_SLICE_GROUPS", "0")
    assert P.compile_tree(tree, "complex64").group_size == 1


def test_slice_groups_on_every_small_sliced_golden_tree(groups_on_small_trees):
    """All sliced golden trees the interpreter can run in full (hyper and output indices sliced, extents 2
    and 3, single-term preprocessing; 77 of them have a step that is independent of some sliced index):
    with slice groups forced on, the plan passes the C ABI's validation and the interpreter -- group by
    group, shared steps once per group -- lands on the oracle's contraction."""
    import golden_util as G
    from cotengra_amd import plan as P, runtime
    from oracle import plan_interp

    n = 0
    for case in G.cases("tree"):
        tree = G.tree_of(case)
        if tree.multiplicity < 4 or tree.multiplicity > 4096 or tree.max_size() > 2**16:
            continue
        plan = P.compile_tree(tree, "complex128")
        if plan.group_size < 2:
            continue
        runtime.DevicePlan(plan).close()
        arrays = G.arrays_of(case, "complex128", tree)
        got = plan_interp.run_plan(plan, arrays)
        ref = np.asarray(orc.contract(tree, arrays))
        assert np.allclose(got, ref, rtol=1e-10, atol=1e-12 * np.abs(ref).max()), case["name"]
        n += 1
    assert n >= 60


def test_bench_deals_whole_slice_groups_to_ranks():
    """bench.py's timed slices are whole slice groups, dealt round-robin to the ranks; the flops it counts
    are the ones executed (a shared step once per group)."""
    import importlib.util

    from cotengra_amd import plan as P

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    tree = ca.tree_from_record(ca.load_network(os.path.join(ROOT, "tests", "golden", "trees", "sycamore_m20_w32_r4.json")))
    os.environ.pop("CTG_SLICE_GROUPS", None)
    plan = P.compile_tree(tree, "complex64")
    gs, world = plan.group_size, 8
    assert gs == 4
    seen = []
    for rank in range(world):
        ids = bench.slice_ids_from_groups(plan, 0, 2 * gs, rank, world)
        assert len(ids) == 2 * gs and [plan.group_of(i) for i in ids] == [rank] * gs + [rank + world] * gs
        seen += ids
    assert len(set(seen)) == len(seen) and all(0 <= i < plan.nslices for i in seen)
    ids = bench.slice_ids_from_groups(plan, 0, gs + 1)
    full, part = bench.executed_flops(plan, ids[:gs]), bench.executed_flops(plan, ids)
    nominal = 8.0 * plan.macs_per_slice
    assert full == pytest.approx(gs * nominal - 8.0 * plan.macs_shared_per_group * (gs - 1))
    assert part - full == pytest.approx(nominal)            # (the lone slice of the next group pays for everything)
    note = bench.slice_groups_note(plan, ids)
    assert note["slices_per_group"] == gs and note["timed"] == "%d slices in 2 groups" % (gs + 1)
