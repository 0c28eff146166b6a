"""CPU suite, part 5 (round 2): host logic added with C ABI v2.

* projected OUTPUT indices keep a size-1 axis (golden from the reference);
* projections survive every path that rebuilds a tree;
* (the ordered traversal moved to test_host_round3.py: reference-frozen sequences);
* the per-op plug-in ``implementation=(einsum, tensordot)`` (contract.py:775-776)
  driven with numpy's functions against the oracle;
* checkpoint files: signature, atomic replace, mismatch;
* the collective entry points: id creation and argument validation without a GPU;
* ``bench.py --gpus N`` starts its own ranks.
"""
import os
import subprocess
import sys
import types

import numpy as np
import pytest

import cotengra_amd as ca
from cotengra_amd import runtime
from cotengra_amd.contractor import (
    PerOpContractor, load_checkpoint, save_checkpoint, tree_signature,
)
from cotengra_amd.plan import compile_tree
from oracle import contract_ref as orc
from oracle.plan_interp import run_plan

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R2_CASES, R2_EXPECTED = G.load_r2()


@pytest.mark.parametrize("case", R2_CASES, ids=[c["name"] for c in R2_CASES])
def test_projected_output_index_keeps_unit_axis(case):
    tree = G.tree_of(case)
    arrays = G.arrays_of(case, "complex128", tree)
    ref = R2_EXPECTED[f"{case['name']}/complex128"]
    assert tree.gathered_shape() == ref.shape and 1 in ref.shape
    assert tree.nslices == case["stats"]["nslices"]
    assert G.relerr(orc.contract(tree, arrays), ref) < 1e-12
    plan = compile_tree(tree, "complex128")
    assert tuple(plan.result_shape) == ref.shape
    assert G.relerr(run_plan(plan, arrays), ref) < 1e-12
    for i in case["slice_ids"]:
        sl = R2_EXPECTED[f"{case['name']}/complex128/slice{i}"]
        assert G.relerr(orc.contract_slice(tree, arrays, i), sl) < 1e-12


def _projected_tree():
    inputs, output, shapes, size_dict = ca.lattice_equation([3, 3], d_min=2, d_max=4, seed=2)
    tree = ca.ContractionTree.from_path(inputs, output, size_dict, path=ca.greedy_path(inputs, output, size_dict))
    tree.remove_ind_(inputs[4][0], project=1)
    tree.remove_ind_(inputs[0][0])
    return tree


def test_projection_survives_tree_rebuilds():
    tree = _projected_tree()
    proj = {ix: si.project for ix, si in tree.sliced_inds.items()}
    assert sorted(v for v in proj.values() if v is not None) == [1]
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=3)
    ref = orc.contract(tree, arrays)
    # native subtree reconfiguration rebuilds the tree from an SSA path
    new = tree.subtree_reconfigure(subtree_size=4)
    assert {ix: si.project for ix, si in new.sliced_inds.items()} == proj
    assert new.nslices == tree.nslices
    assert np.allclose(orc.contract(new, arrays), ref, rtol=1e-12, atol=1e-15)
    # record round trip: [name, j] entries
    rec = {"inputs": tree.inputs, "output": tree.output, "size_dict": tree.size_dict,
           "path": tree.get_path(), "sliced_inds": tree.slicing_record()}
    assert any(isinstance(e, list) for e in rec["sliced_inds"])
    back = ca.tree_from_record(rec)
    assert {ix: si.project for ix, si in back.sliced_inds.items()} == proj
    # adopting a foreign tree object (anything with get_path / sliced_inds: the reference's)
    foreign = types.SimpleNamespace(
        inputs=tree.inputs, output=tree.output, size_dict=tree.size_dict,
        get_path=tree.get_path, sliced_inds=dict(tree.sliced_inds),
    )
    adopted = ca.array_contract_tree(tree.inputs, tree.output, tree.size_dict, optimize=foreign)
    assert {ix: si.project for ix, si in adopted.sliced_inds.items()} == proj
    assert np.allclose(orc.contract(adopted, arrays), ref, rtol=1e-12, atol=1e-15)


def _np_pair():
    return (np.einsum, np.tensordot)


@pytest.mark.parametrize(
    "name", ["lattice4x4_sliced", "rand_s42_r2_o2_hi1_ho2_outsliced", "preproc_s3", "C1_rand10_d4"],
)
def test_per_op_plugin_with_numpy_functions(name):
    cands = [c for c in G.cases("tree") if c["name"] == name]
    if not cands:
        cands = [c for c in G.cases("tree") if c["name"].startswith(name.split("_")[0])][:1]
    case = cands[0]
    tree = G.tree_of(case)
    dt = case["dtypes"][0]
    arrays = G.arrays_of(case, dt, tree)
    if case["slice_ids"]:
        i = case["slice_ids"][0]
        got = tree.contract_slice(arrays, i, implementation=_np_pair())
        assert G.relerr(got, G.expected(f"{case['name']}/{dt}/slice{i}")) < 1e-11
        return
    ref = G.expected(f"{case['name']}/{dt}")
    got = tree.contract(arrays, implementation=_np_pair())
    assert G.relerr(got, ref) < 1e-11
    m, e = tree.contract(arrays, implementation=_np_pair(), strip_exponent=True)
    assert G.relerr(np.asarray(m) * 10.0**e, ref) < 1e-11
    got = tree.contract(arrays, implementation=_np_pair(), prefer_einsum=True)
    assert G.relerr(got, ref) < 1e-11


def test_per_op_plugin_schedule_and_errors():
    tree = ca.ContractionTree.from_path(["ab", "bc", "cd"], "ad", dict(a=3, b=4, c=5, d=2), path=[(0, 1), (0, 1)])
    calls = []

    def einsum(eq, *xs):
        calls.append(("einsum", eq))
        return np.einsum(eq, *xs)

    def tensordot(a, b, axes):
        calls.append(("tensordot", axes))
        return np.tensordot(a, b, axes)

    xs = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=0)
    out = tree.contract(xs, implementation=(einsum, tensordot))
    assert np.allclose(out, np.einsum("ab,bc,cd->ad", *xs))
    assert [c[0] for c in calls] == ["tensordot", "tensordot"]  # (einsum, tensordot) order, contract.py:775
    fn = tree.get_contractor(implementation=(einsum, tensordot))
    assert isinstance(fn, PerOpContractor) and fn is tree.get_contractor(implementation=(einsum, tensordot))
    with pytest.raises(TypeError):
        fn(*xs, no_such_option=1)
    with pytest.raises(ValueError):
        fn(*xs[:2])
    with pytest.raises(ValueError):
        tree.get_contractor(implementation=(einsum,))
    with pytest.raises(ValueError):
        tree.get_contractor(implementation="cuquantum")
    xs[1] = np.zeros_like(xs[1])
    assert tree.contract(xs, implementation=(einsum, tensordot), strip_exponent=True, check_zero=True) == (
        0.0, float("-inf"))
    # single-input tree: one einsum call
    t1 = ca.ContractionTree(["aab"], "b", dict(a=3, b=4))
    x = np.arange(36.0).reshape(3, 3, 4)
    assert np.allclose(t1.contract([x], implementation=_np_pair()), np.einsum("aab->b", x))


def test_checkpoint_files(tmp_path):
    tree = _projected_tree()
    sig = tree_signature(tree, "complex128")
    assert sig == tree_signature(tree.copy(), "complex128")
    assert sig != tree_signature(tree, "complex64")
    assert sig != tree_signature(tree, "complex128", rank=1, world=2)
    assert sig != tree_signature(tree, "complex128", strip_exponent=True)
    assert sig != tree_signature(tree.restore_ind(next(iter(tree.sliced_inds))), "complex128")
    path = str(tmp_path / "run.ckpt")
    assert load_checkpoint(path, sig) is None
    part = np.arange(6, dtype=np.complex128).reshape(2, 3) * (1 - 2j)
    save_checkpoint(path, sig, 3, part, exponent=-4.5, zero=False)
    done, res, e, z = load_checkpoint(path, sig)
    assert (done, e, z) == (3, -4.5, False) and np.array_equal(res, part) and res.dtype == part.dtype
    save_checkpoint(path, sig, 5, part * 2)  # replaces atomically, no stray temp files
    assert load_checkpoint(path, sig)[0] == 5
    assert os.listdir(tmp_path) == ["run.ckpt"]
    with pytest.raises(ValueError, match="different"):
        load_checkpoint(path, tree_signature(tree, "complex64"))


def test_collective_entry_points_without_gpu():
    uid = runtime.Comm.unique_id()
    assert len(uid) == 128 and uid != runtime.Comm.unique_id()
    with pytest.raises(ValueError):
        runtime.Comm(uid, rank=2, world=2)  # rank outside the world: CTG_E_INVALID before any device call
    with pytest.raises(ValueError):
        runtime.Comm(b"short", 0, 1)
    assert issubclass(runtime.CommError, runtime.CtgError)


def test_bench_starts_its_own_ranks(monkeypatch):
    """``python bench.py --gpus 4`` with no launcher in the environment must
    re-execute itself under torch.distributed.run with 4 ranks on 127.0.0.1."""
    sys.path.insert(0, ROOT)
    import importlib

    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_full_device_evicts_sibling_executors(monkeypatch):
    """A tree's cached contractors each keep an arena resident; when a new
    executor does not fit, the siblings' executors are closed and the
    allocation retried (they are rebuilt on demand)."""
    from cotengra_amd import contractor as ctr

    tree = _projected_tree()
    closed = []

    class FakeExec:
        def __init__(self, tag):
            self.tag = tag

        def close(self):
            closed.append(self.tag)

    attempts = []

    def fake_executor(dplan, device=0, stream=0, result_ptr=None):
        attempts.append(device)
        if len(attempts) == 1:
            raise MemoryError("hipMalloc failed: out of memory")
        return FakeExec("new")

    monkeypatch.setattr(ctr.runtime, "Executor", fake_executor)
    monkeypatch.setattr(ctr.runtime, "DevicePlan", lambda plan: object())
    sibling = ctr.HipContractor(tree, handle_slicing=False)
    sibling._execs[("complex128", 0, False)] = {"exec": FakeExec("sibling"), "plan": None}
    tree.contraction_cores["some-key"] = sibling
    fn = ctr.HipContractor(tree, handle_slicing=True)
    st = fn._get_exec("complex128", 0, False)
    assert st["exec"].tag == "new" and closed == ["sibling"] and len(attempts) == 2
    assert sibling._execs == {}
    # nothing to evict: the error surfaces
    attempts.clear()
    fn2 = ctr.HipContractor(_projected_tree(), handle_slicing=True)
    with pytest.raises(MemoryError):
        fn2._get_exec("complex64", 0, False)


def test_committed_bench_line_follows_the_contract():
    """The bench line the round published (profiles/r<N>_bench_line.json, written by
    bench.py on the GPU box) carries every field the driver and the judge read."""
    import json

    path = next((q for q in (os.path.join(ROOT, "profiles", f"r{n}_bench_line.json") for n in (5, 4, 3, 2))
                 if os.path.exists(q)), None)
    if path is None:
        pytest.skip("no published bench line")
    d = json.load(open(path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["achieved"] <= r["peak"]
    # (fp32-equivalent flops: with bf16 x 3 pairs the bound is bf16 peak / 6 = 416.7 TF, else the fp32 peak)
    assert d["value"] / 1e12 <= (416.7 if "bf16" in d["dtype"] else 157.3) * d["n_gpus"]
    # the dominant kernel's launches cannot take longer than the step they are part of
    assert r["avg_launch_ms"] * r["launches_per_slice"] <= d["ms_per_step"] * 1.02
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
    # the value is slices x algorithmic flops / time
    assert abs(d["value"] - d["config"]["flops_per_slice"] * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-6 * d["value"]
    assert d["precision"]["rel_err"] <= d["precision"]["gate"]
    # (round 3 on: the headline IS the time-to-solution tree, the former headline rides along)
    t = d["peak_rate_tree"] if "peak_rate_tree" in d else d["time_to_solution_tree"]
    # (round 4 on: the bound is the mixed per-step sum over MOVED bytes, "mixed_bound_frac")
    mixed = t["mixed_bound_frac"] if "mixed_bound_frac" in t else t["mixed_roofline_frac"]
    assert 0 < mixed <= 1
    assert 0 < (t["frac_of_mfma_peak"] if "frac_of_mfma_peak" in t else t["ratio_to_fp32_mfma_peak"]) <= 1
    for name in ("C2", "C3", "C5"):
        cfg = d["configs"][name]
        assert cfg["ms"] > 0 and 0 < cfg["mixed_roofline_frac"] <= 1 and cfg["cpu_oracle_ms"] > 0
    if os.path.basename(path) >= "r4":
        # a fraction of a bound is not above 1 -- anywhere in the line (the fused pairs had
        # outgrown the unfused-bytes accounting; 1 % of timing noise allowed)
        def walk(x, where):
            if isinstance(x, dict):
                for k, v in x.items():
                    if isinstance(v, (int, float)) and "frac" in k and not isinstance(v, bool):
                        assert v <= 1.01, (where + "/" + k, v)
                    walk(v, where + "/" + k)
            elif isinstance(x, list):
                for i, v in enumerate(x):
                    walk(v, f"{where}[{i}]")

        walk(d, "")
        mp = d["roofline"]["mixed_per_step"]
        assert mp["bound_ms"] <= d["ms_per_step"] * 1.01 and mp["bound_ms"] <= mp["unfused_roofline_ms"] * 1.0001
        assert d["roofline"]["moved_bytes_per_launch"] <= d["roofline"]["algorithmic_bytes_per_launch"]


# ---------------------------------------------------------------------- #
# wave-front planning (the executor's build_groups relies on these properties)
# ---------------------------------------------------------------------- #


def _memory_independent(x, y):
    """No read-after-write, write-after-read or write-after-write between two steps
    (the interval test ctg_runtime.hip:build_groups applies)."""

    def iv(t):
        return (t.space, t.offset, t.offset + t.size)

    def ov(p, q):
        return p[0] == q[0] and p[1] < q[2] and q[1] < p[2]

    return not (
        ov(iv(y.a), iv(x.c)) or ov(iv(y.b), iv(x.c)) or ov(iv(y.c), iv(x.a)) or ov(iv(y.c), iv(x.b))
        or ov(iv(y.c), iv(x.c))
    )


@pytest.mark.parametrize("name", ["C2_lattice8x8_d4", "C5_hyper200", "lattice8x8_sliced"])
def test_small_trees_are_planned_level_by_level(name):
    """A small tree is emitted level by level (level = 1 + the deeper child) and
    operands are recycled only when their level is done: all pair steps of one
    level are pairwise independent in memory, so the executor may send them out
    in shared launches.  The depth-first order stays available, and both orders
    hold the same steps."""
    import golden_util as G
    from cotengra_amd import plan as P

    case = next(c for c in G.cases("tree") if c["name"] == name)
    tree = G.tree_of(case)
    level = {}
    for p, l, r in tree.traverse():
        level[p] = 1 + max(level.get(l, 0), level.get(r, 0))
    pl = P.compile_tree(tree, "complex64")
    pairs = [s for s in pl.steps if s.kind == P.KIND_PAIR]
    # (round 6: the members of LDS-resident subtrees come first among the steps of their sharing class --
    # cotengra_amd/ldsrun.py --, level by level; then the other steps of the class, level by level)
    from cotengra_amd.ldsrun import CLASS_RANK

    def section(s):
        return (CLASS_RANK["inv" if s.invariant else ("group" if s.group else "slice")], s.lds_comp < 0)

    seq = [(section(s), level[s.node]) for s in pairs]
    assert seq == sorted(seq), "steps are not in level order"
    by_level = {}
    for s in pairs:
        by_level.setdefault((section(s), level[s.node]), []).append(s)
    for steps in by_level.values():
        for i, x in enumerate(steps):
            for y in steps[i + 1:]:
                assert _memory_independent(x, y), (x.label, y.label)
    dfs = P.compile_tree(tree, "complex64", order="dfs")
    assert sorted(s.node for s in dfs.steps) == sorted(s.node for s in pl.steps)
    assert pl.macs_per_slice == dfs.macs_per_slice
    assert dfs.arena_elems <= pl.arena_elems <= 2 * dfs.arena_elems
    # consecutive independent thread-per-output steps form the shared launches
    launches, i = 0, 0
    live = [s for s in pl.steps if not s.invariant]
    groupable = lambda s: (s.kind == P.KIND_PAIR and s.kernel == P.KERNEL_VALU  # noqa: E731
                           and not (s.K >= 256 and s.R * s.N <= 32768))
    while i < len(live):
        j = i + 1
        if groupable(live[i]):
            while j < len(live) and groupable(live[j]) and all(_memory_independent(live[q], live[j]) for q in range(i, j)):
                j += 1
        launches += 1
        i = j
    if name.startswith("C"):
        assert launches * 2 <= len(live), (launches, len(live))


def test_wide_trees_keep_the_depth_first_order():
    """Level order is for trees whose largest intermediate is small; a wide tree
    keeps the reference's depth-first order (and its smaller peak memory)."""
    import cotengra_amd as ca
    from cotengra_amd import plan as P

    rec = ca.load_network(os.path.join(ROOT, "tests", "golden", "trees", "sycamore_m20_w30.json"))
    tree = ca.tree_from_record(rec)
    assert tree.max_size() > P.LEVEL_ORDER_MAX_ELEMS
    a = P.compile_tree(tree, "complex64")
    b = P.compile_tree(tree, "complex64", order="dfs")
    assert [s.node for s in a.steps] == [s.node for s in b.steps]
    assert a.arena_elems == b.arena_elems
