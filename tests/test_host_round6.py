"""Round 6, host side: the LDS-resident subtrees (cotengra_amd/ldsrun.py) -- what the planner cuts off,
that the second lowering computes what the ordinary steps compute (numpy interpreter of both), and that
the C ABI refuses a component descriptor that would reach outside its buffers.  No GPU."""
import numpy as np
import pytest

import golden_util as G
from cotengra_amd import ldsrun, plan as P, runtime
from oracle import contract_ref as orc
from oracle import plan_interp as PI

SMALL = [c for c in G.cases("tree") if not c["name"].startswith(("C4_", "C5_", "C2_"))]


def _plan(case, dtype):
    tree = G.tree_of(case)
    return tree, P.compile_tree(tree, dtype)


def _class(step):
    return "inv" if step.invariant else ("group" if step.group else "slice")


@pytest.mark.parametrize("name", ["C2_lattice8x8_d4", "C5_hyper200", "C1_rand10_d4"])
def test_components_are_closed_subtrees_that_fit(name):
    case = next(c for c in G.cases("tree") if c["name"] == name)
    tree, plan = _plan(case, "complex64")
    assert plan.lds_runs, "no LDS-resident subtree found"
    cap = ldsrun.LDS_DATA_BYTES // plan.itemsize
    producer = {id(s.c): i for i, s in enumerate(plan.steps) if s.kind != P.KIND_ACCUM}
    seen = set()
    for run in plan.lds_runs:
        members = set(run["members"])
        assert not (members & seen)
        seen |= members
        assert 0 < run["lds_elems"] <= cap
        classes = {_class(plan.steps[m]) for m in members}
        assert classes == {run["cls"]} and run["cls"] in ("group", "slice")
        for m in members:
            st = plan.steps[m]
            assert st.lds_comp == run["id"] and st.kind in (P.KIND_PAIR, P.KIND_SINGLE)
            for op in (st.a, st.b):
                if op is None or id(op) not in producer:
                    continue   # a leaf view
                w = producer[id(op)]
                # closed: an operand is made by a member, or by a step of a class that runs less often
                assert w in members or ldsrun.CLASS_RANK[_class(plan.steps[w])] < ldsrun.CLASS_RANK[run["cls"]], (m, w)
        # exactly one result leaves the component, and no other member's result is read outside
        for m in members:
            readers = [i for i, s in enumerate(plan.steps) if s.kind != P.KIND_ACCUM and (s.a is plan.steps[m].c or s.b is plan.steps[m].c)]
            if m == run["root_main"]:
                assert all(r not in members for r in readers)
            else:
                assert readers and all(r in members for r in readers), (m, readers)
        # LDS tensors stay inside the data area
        for sh in run["steps"]:
            for t in (sh["step"].a, sh["step"].b, sh["step"].c):
                if t is not None and t.space == ldsrun.SPACE_LDS:
                    assert 0 <= t.offset and t.offset + t.size <= run["lds_elems"]


@pytest.mark.parametrize("name", ["C2_lattice8x8_d4", "C5_hyper200", "C1_rand10_d4", "preproc_s0_a"])
def test_members_come_first_within_their_class(name):
    """The run of a class is launched where its first member stands: everything a member reads must be
    complete by then, i.e. made by an earlier class or by a member -- and no step of the class that is not
    a member may stand before a member (the arena offsets are assigned in step order)."""
    case = next(c for c in G.cases("tree") if c["name"] == name)
    _, plan = _plan(case, "complex64")
    rank = ldsrun.CLASS_RANK
    last = {}
    for i, s in enumerate(plan.steps):
        if s.kind == P.KIND_ACCUM:
            continue
        key = (rank[_class(s)], s.lds_comp < 0)
        kind = "single" if s.kind == P.KIND_SINGLE else "pair"
        assert last.get(kind, key) <= key, (name, i, kind, key)
        last[kind] = key


@pytest.mark.parametrize("dtype", ["complex128", "float64"])
def test_shadow_lowering_equals_the_ordinary_steps(dtype):
    """Every golden tree small enough for the numpy interpreter: the plan run through its LDS components
    equals the plan run step by step, bit for bit (same sums in the same order), and the oracle."""
    ran = 0
    for case in SMALL:
        tree = G.tree_of(case)
        if tree.N < 4 or tree.max_size() > (1 << 16):
            continue
        if dtype not in case["dtypes"]:
            continue
        plan = P.compile_tree(tree, dtype)
        if not plan.lds_runs:
            continue
        arrays = G.arrays_of(case, dtype, tree)
        ids = list(range(min(plan.nslices, 8)))
        a = PI.run_plan(plan, arrays, slice_ids=ids, lds=False)
        b = PI.run_plan(plan, arrays, slice_ids=ids, lds=True)
        assert np.array_equal(a, b) or np.allclose(a, b, rtol=1e-13, atol=0), case["name"]
        if plan.nslices <= 8:
            ref = orc.contract(tree, arrays)
            assert np.allclose(b, ref, rtol=1e-10, atol=1e-300), case["name"]
        ran += 1
    assert ran >= (20 if dtype == "complex128" else 3), ran


@pytest.mark.parametrize("extra", [(), ("e",), ("q",)])
def test_per_slice_subtree_reads_what_a_slice_group_shares(extra, monkeypatch):
    """A per-slice component whose first member is a leaf's preprocessing step (which stands before ALL pair
    steps) and which reads a group-shared result: the component runs where its first member PAIR stands,
    behind the steps its group shares."""
    from test_host_round5 import shared_single_tree
    import cotengra_amd as ca

    monkeypatch.delenv("CTG_SLICE_GROUPS", raising=False)
    tree = shared_single_tree(extra)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=11, dtype="complex128")
    plan = P.compile_tree(tree, "complex128")
    assert plan.lds_runs and plan.group_size >= 2
    a = PI.run_plan(plan, arrays, lds=False)
    b = PI.run_plan(plan, arrays, lds=True)
    ref = orc.contract(tree, arrays)
    assert np.allclose(a, ref, rtol=1e-10) and np.array_equal(a, b)
    runtime.DevicePlan(plan).close()


def test_lds_runs_can_be_switched_off(monkeypatch):
    case = next(c for c in G.cases("tree") if c["name"] == "C1_rand10_d4")
    monkeypatch.setenv("CTG_LDS_RUNS", "0")
    _, plan = _plan(case, "complex64")
    assert not plan.lds_runs and all(s.lds_comp < 0 for s in plan.steps)
    monkeypatch.delenv("CTG_LDS_RUNS")
    _, plan = _plan(case, "complex64")
    assert plan.lds_runs


class _Tampered:
    """A plan whose serialised form was changed after the fact (what comes through the C ABI is not trusted)."""

    def __init__(self, plan, ser):
        self._ser = ser
        self.inputs_elems = plan.inputs_elems

    def serialise(self):
        return self._ser


def _first_run(ser):
    steps = ser["steps"].reshape(-1, P.STEP_WORDS)
    m = int(np.nonzero(steps[:, P.W_LDS_COMP] > 0)[0][0])
    return steps, m, int(steps[m, P.W_LDS_DESC])


def test_c_abi_accepts_the_planner_and_refuses_bad_descriptors():
    case = next(c for c in G.cases("tree") if c["name"] == "C1_rand10_d4")
    _, plan = _plan(case, "complex64")
    runtime.DevicePlan(plan).close()
    H, W = ldsrun.LR_HEAD_WORDS, ldsrun.LR_WORDS

    def tampered(change):
        ser = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in plan.serialise().items()}
        steps, m, d = _first_run(ser)
        change(ser, steps, m, d)
        ser["steps"] = steps.reshape(-1)
        return _Tampered(plan, ser)

    def expect_refused(change, what):
        with pytest.raises(Exception) as err:
            runtime.DevicePlan(tampered(change)).close()
        assert what in str(err.value), str(err.value)

    # magic
    expect_refused(lambda s, st, m, d: s["tables"].__setitem__(d, 7), "bad descriptor")
    # data area smaller than what the records address
    expect_refused(lambda s, st, m, d: s["tables"].__setitem__(d + 2, 1), "reaches LDS element")
    # an LDS offset pushed outside the data area
    def push_c(s, st, m, d):
        n = int(s["tables"][d + 1])
        for i in range(n):
            q = d + H + i * W
            if s["tables"][q + 8] == 1:
                s["tables"][q + 9] += 1 << 20
                return
    expect_refused(push_c, "reaches LDS element")
    # a record that claims a step which is not a member
    def foreign(s, st, m, d):
        other = int(np.nonzero(st[:, P.W_LDS_COMP] == 0)[0][0])
        s["tables"][d + H + 2] = other
    expect_refused(foreign, "not a member")
    # a table pointer outside the blob
    expect_refused(lambda s, st, m, d: s["tables"].__setitem__(d + H + 17, len(s["tables"]) + 5), "outside the blob")
    # a member without a record: drop the last record
    def drop(s, st, m, d):
        s["tables"][d + 1] -= 1
    expect_refused(drop, "has no record")
    # descriptor pointer outside the blob
    expect_refused(lambda s, st, m, d: st.__setitem__((m, P.W_LDS_DESC), len(s["tables"])), "outside the blob")
