"""CPU suite, round 5: how slices are dealt to ranks (whole slice groups: Plan.share_units /
ctg_plan_share_units / ctg_exec_run_share), on the host side of the library and through gloo."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cotengra_amd as ca  # noqa: E402

TREES = os.path.join(ROOT, "tests", "golden", "trees")
M20 = ["sycamore_m20_native.json", "sycamore_m20_w32_r4.json", "sycamore_m20_w33_bf3.json", "sycamore_m20_w32_g.json",
       "sycamore_m20_w32_c512.json"]

# the advisor's round-4 tree: a leaf that is preprocessed (index z summed away) depends on the sliced
# index a but not on the group index g, and feeds a step that depends on g
SHARED_SINGLE = dict(
    inputs=[("a", "b", "z"), ("b", "g", "c"), ("c", "a", "g", "d"), ("d", "a", "e", "f"), ("e", "f", "q", "r"),
            ("q", "r", "s"), ("s",)],
    sizes={**{k: 2 for k in "abzgcds"}, **{k: 8 for k in "efqr"}},
    ssa_path=[(4, 5), (7, 6), (3, 8), (0, 1), (10, 2), (11, 9)],
    sliced=("a", "g"),
)


def shared_single_tree(extra_sliced=()):
    t = ca.ContractionTree.from_path(SHARED_SINGLE["inputs"], (), SHARED_SINGLE["sizes"],
                                     ssa_path=SHARED_SINGLE["ssa_path"])
    for ix in SHARED_SINGLE["sliced"] + tuple(extra_sliced):
        t.remove_ind_(ix)
    return t


@pytest.mark.parametrize("fixture", M20)
def test_ranks_hold_whole_slice_groups(fixture):
    """World 1 / 2 / 3 / 4 / 8 on the m20 fixtures: every rank holds only complete groups, the shares are
    disjoint, cover range(nslices) and differ by at most one group; the C library enumerates the same ids."""
    from collections import Counter

    from cotengra_amd.contractor import _tree_contractor

    tree = ca.tree_from_record(ca.load_network(os.path.join(TREES, fixture)))
    plan, dplan = _tree_contractor(tree, None).get_plan("complex64")
    gs = int(plan.group_size)
    assert gs >= 2
    for world in (1, 2, 3, 4, 8):
        shares = [plan.rank_slice_ids(r, world) for r in range(world)]
        for r, ids in enumerate(shares):
            units, per = plan.share_units(r, world)
            assert (units, per) == dplan.share_units(r, world) and per == gs and len(ids) == units * gs
            members = Counter(np.asarray(plan.group_of(ids)).tolist())
            assert set(members.values()) == {gs}                       # only complete groups
            assert sorted(members) == list(range(r, tree.nslices // gs, world))[: len(members)]
            # the library's enumeration, head, tail and a window in the middle
            for u0, n in ((0, min(units, 7)), (max(units - 5, 0), -1), (units // 2, min(3, units - units // 2))):
                c = dplan.share_slice_ids(r, world, u0, n)
                assert np.array_equal(c, ids[u0 * gs: u0 * gs + len(c)])
        sizes = [len(s) for s in shares]
        assert max(sizes) - min(sizes) <= gs
        allids = np.concatenate(shares)
        assert len(allids) == tree.nslices and len(np.unique(allids)) == tree.nslices
        assert allids.min() == 0 and allids.max() == tree.nslices - 1


def test_share_without_groups_is_the_reference_round_robin(monkeypatch):
    """No group indices in the plan: unit = slice, the share is range(rank, nslices, world) (core.py:4070)."""
    from cotengra_amd import plan as P, runtime

    monkeypatch.setenv("CTG_SLICE_GROUPS", "0")
    tree = shared_single_tree()
    plan = P.compile_tree(tree, "complex64")
    assert plan.group_size == 1
    d = runtime.DevicePlan(plan)
    for world in (1, 2, 3, 4):
        for r in range(world):
            want = list(range(r, tree.nslices, world))
            assert plan.rank_slice_ids(r, world).tolist() == want
            assert d.share_slice_ids(r, world).tolist() == want
            assert d.share_units(r, world) == (len(want), 1)
    with pytest.raises(ValueError):
        d.share_units(2, 2)
    with pytest.raises(ValueError):
        d.share_slice_ids(0, 2, 1, 5)
    d.close()


def test_share_on_small_grouped_golden_trees(monkeypatch):
    """Group extents 2 and 3, projected indices, more ranks than groups: C and Python agree, shares are whole
    groups and partition the slices."""
    import golden_util as G
    from cotengra_amd import plan as P, runtime

    monkeypatch.delenv("CTG_SLICE_GROUPS", raising=False)
    monkeypatch.setattr(P, "GROUP_MIN_WIDTH", 1)
    monkeypatch.setattr(P, "GROUP_MIN_SAVING", 0.0)
    n = 0
    for case in G.cases("tree"):
        tree = G.tree_of(case)
        if tree.multiplicity < 4 or tree.multiplicity > 4096:
            continue
        plan = P.compile_tree(tree, "complex128")
        if plan.group_size < 2:
            continue
        d = runtime.DevicePlan(plan)
        gs = int(plan.group_size)
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                ids = d.share_slice_ids(r, world)
                assert np.array_equal(ids, plan.rank_slice_ids(r, world))
                for u in range(len(ids) // gs):
                    assert ids[u * gs:(u + 1) * gs].tolist() == plan.group_ids(r + u * world)
                got += ids.tolist()
            assert sorted(got) == list(range(tree.multiplicity))
        d.close()
        n += 1
    assert n >= 60


def test_shared_single_step_plan_is_what_the_advisor_described(monkeypatch):
    """The plan of the round-4 finding: a SHARED leaf-preprocessing step (kind 0) whose consumer is a
    per-slice pair step that names no producer (a_prod / b_prod cover pair and stem producers only).  The
    executor finds the writer of an operand in the arena (csrc/ctg_runtime.hip: resolve_args); the
    device run is tests/test_gpu_round5.py."""
    from cotengra_amd import plan as P, runtime
    from oracle import contract_ref as orc, plan_interp

    monkeypatch.delenv("CTG_SLICE_GROUPS", raising=False)
    tree = shared_single_tree()
    plan = P.compile_tree(tree, "complex64")
    assert plan.group_inds == ("g",) and plan.group_size == 2
    single = plan.steps[0]
    assert single.kind == P.KIND_SINGLE and single.group and not single.invariant
    consumer = next(s for s in plan.steps if s.kind == P.KIND_PAIR and (s.a is single.c or s.b is single.c))
    assert not consumer.group and consumer.a_prod == -1 and consumer.b_prod == -1
    runtime.DevicePlan(plan).close()
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=3, dtype="complex128")
    plan128 = P.compile_tree(tree, "complex128")
    got = complex(np.asarray(plan_interp.run_plan(plan128, arrays)))
    ref = complex(orc.contract(tree, arrays))
    assert abs(got - ref) <= 1e-10 * abs(ref)


# ---- gloo, world 2: the injected executor is dealt whole groups ----------------------------------------------


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from cotengra_amd import plan as P
    from cotengra_amd.distributed import contract_distributed
    from oracle import contract_ref as orc
    from test_host_round5 import shared_single_tree

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("CTG_SLICE_GROUPS", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tree = shared_single_tree(extra_sliced=("e",))          # 2 x 2 x 8 = 32 slices, groups of 4 (a, g)
        arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=5, dtype="complex128")
        plan = P.compile_tree(tree, "complex128")
        seen = {}

        def executor(tree_, arrays_, mine):
            seen["mine"] = list(mine)
            return sum(orc.contract_slice(tree_, arrays_, i) for i in mine)

        out = contract_distributed(tree, arrays, executor_factory=executor)
        ok = bool(np.allclose(out.numpy(), orc.contract(tree, arrays), rtol=1e-12))
        mine = seen["mine"]
        gs = int(plan.group_size)
        ok = ok and gs == 4 and mine == plan.rank_slice_ids(rank, world).tolist()
        # only whole groups, the groups rank, rank + world, ...
        groups = [plan.group_of(i) for i in mine]
        ok = ok and groups == [g for g in range(rank, tree.nslices // gs, world) for _ in range(gs)]
        q.put((rank, ok, len(mine)))
    finally:
        dist.destroy_process_group()


def test_whole_groups_through_gloo_world2():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
    results = sorted(q.get(timeout=10) for _ in procs)
    assert results == [(0, True, 16), (1, True, 16)]
    assert all(p.exitcode == 0 for p in procs)
