"""GPU suite, round 3.

* ``contract(order=f)`` follows the reference's ordered traversal and gives the very bits of
  the default order (the schedule changes lifetimes, not values);
"""
import json
import os

import numpy as np
import pytest

import cotengra_amd as ca
from oracle import contract_ref as orc

import golden_util as G
from test_host_round3 import ORDERS, TRAVERSE, traverse_tree

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["lattice4x4_s0", "hyper24_s0", "randreg30_s4", "lattice3x3x3_s2"])
@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
def test_ordered_contract_same_bits(name, dtype):
    case = next(c for c in TRAVERSE if c["name"] == name)
    tree = traverse_tree(case)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=3, dtype=dtype, rescale=True)
    ref = orc.contract(tree, [a.astype("complex128") for a in arrays])
    base = np.asarray(tree.contract(arrays))
    tol = 1e-10 if dtype == "complex128" else G.single_gate(ref, orc.contract(tree, arrays))
    assert G.relerr(base, ref) <= tol
    for oname, make in ORDERS.items():
        got = np.asarray(tree.contract(arrays, order=make(tree)))
        assert got.dtype == base.dtype and np.array_equal(got, base), oname


# ---------------------------------------------------------------------- #
# fused stem pairs: csrc/ctg_stem.hip against the oracle and the unfused path
# ---------------------------------------------------------------------- #

from cotengra_amd.contractor import HipContractor  # noqa: E402
from cotengra_amd.plan import KIND_STEM2  # noqa: E402


@pytest.fixture
def fuse_whatever_fits(monkeypatch):
    """(every pair the kernel can take, whatever the pairing model thinks of its gathers)"""
    from cotengra_amd import stem
    monkeypatch.setattr(stem, "gather_rate", lambda run_bytes: 5.4e12)


@pytest.mark.parametrize("sliced", [0, 2])
@pytest.mark.parametrize("case", range(len(G.STEM_CASES)))
def test_fused_stem_pairs(case, sliced, fuse_whatever_fits):
    nq, gates = G.STEM_CASES[case]
    tree = G.stem_network(nq, gates, 100 * case, sliced=sliced)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=case, dtype="complex64")
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    fused = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    plain = HipContractor(tree, fuse=False)
    plan = fused.get_plan("complex64")[0]
    n_fused = sum(s.kind == KIND_STEM2 for s in plan.steps)
    got = np.asarray(fused(*arrays))
    base = np.asarray(plain(*arrays))
    if n_fused:
        st = fused.setup(*arrays)
        assert sum(n.startswith("stem2_kernel") for n in st["exec"].step_kernels()) == n_fused
    gate = G.single_gate(ref, orc.contract(tree, arrays))
    assert G.relerr(got, ref) <= gate, (G.relerr(got, ref), gate)
    assert G.relerr(base, ref) <= gate
    # strip_exponent: one factor per fused pair, same number
    m, e = fused(*arrays, strip_exponent=True)
    assert G.relerr(np.asarray(m).astype("complex128") * 10.0**e, ref) <= gate
    # slice by slice = all slices in one batched run
    if tree.nslices > 1:
        a128 = [a.astype("complex128") for a in arrays]
        for i in range(tree.nslices):   # (a sliced index may be an open one: compare slice by slice)
            ri = np.asarray(orc.contract_slice(tree, a128, i))
            gi = max(gate, G.single_gate(ri, orc.contract_slice(tree, arrays, i)))
            assert G.relerr(np.asarray(fused.contract_slice(arrays, i)), ri) <= gi
    fused.close()
    plain.close()


@pytest.mark.parametrize("case", range(len(G.STEM_CASES)))
def test_fused_stem_pairs_on_the_bf16_matrix_cores(case, fuse_whatever_fits, monkeypatch):
    """bf16 x 3 (the default arithmetic of the fused pairs since round 4; CTG_STEM_BF16X3=0 = fp32
    products): both steps of a pair multiply on the bf16 matrix cores -- every fp32 operand split
    exactly into three bf16 values, the six significant cross terms accumulated in fp32.  Same gate
    as the fp32 path against the numpy complex128 oracle; shapes without a static instantiation
    keep the fp32 kernel."""
    nq, gates = G.STEM_CASES[case]
    tree = G.stem_network(nq, gates, 100 * case)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=case, dtype="complex64")
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    gate = G.single_gate(ref, orc.contract(tree, arrays))
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    default = np.asarray(fn(*arrays))
    monkeypatch.setenv("CTG_STEM_BF16X3", "0")
    fp32 = np.asarray(fn(*arrays))
    assert not any(n.startswith("stem2_kernel") and G.stem_flags(n)["bf3"] for n in fn.setup(*arrays)["exec"].step_kernels())
    monkeypatch.setenv("CTG_STEM_BF16X3", "1")
    got = np.asarray(fn(*arrays))
    assert np.array_equal(got, default)     # (bf16 x 3 is what runs when nothing is said)
    names = [n for n in fn.setup(*arrays)["exec"].step_kernels() if n.startswith("stem2_kernel")]
    m, e = fn(*arrays, strip_exponent=True)
    fn.close()
    assert names
    if any(G.stem_flags(n)["bf3"] for n in names):   # (the tenth template argument: BF3)
        assert not np.array_equal(got, fp32)   # (it really ran)
    assert G.relerr(got, ref) <= gate, (G.relerr(got, ref), gate, G.relerr(fp32, ref))
    assert G.relerr(np.asarray(m).astype("complex128") * 10.0**e, ref) <= gate
    # the same through the C ABI's per-executor option instead of the environment: the same bits
    monkeypatch.delenv("CTG_STEM_BF16X3")
    fn2 = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10, stem_bf16x3=True)
    again = np.asarray(fn2(*arrays))
    names2 = [n for n in fn2.setup(*arrays)["exec"].step_kernels() if n.startswith("stem2_kernel")]
    fn2.close()
    assert names2 == names and np.array_equal(again, got)
    fn3 = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10, stem_bf16x3=False)   # ... and off
    assert np.array_equal(np.asarray(fn3(*arrays)), fp32)
    fn3.close()


@pytest.mark.parametrize("seed", range(48))
def test_random_stems_on_the_fused_kernel(seed, fuse_whatever_fits, monkeypatch):
    """The 48 random stems of the host suite (58 fused pairs in 50 shapes, most of them on the
    run-time-count variant) on the device, fp32 matrix cores and bf16 x 3, against the oracle."""
    tree = G.random_stem(seed)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=seed, dtype="complex64")
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    gate = G.single_gate(ref, orc.contract(tree, arrays))
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 9)
    monkeypatch.setenv("CTG_STEM_BF16X3", "0")
    got = np.asarray(fn(*arrays))
    monkeypatch.setenv("CTG_STEM_BF16X3", "1")
    got3 = np.asarray(fn(*arrays))
    fn.close()
    assert G.relerr(got, ref) <= gate, (G.relerr(got, ref), gate)
    assert G.relerr(got3, ref) <= gate, (G.relerr(got3, ref), gate)


@pytest.mark.parametrize("case", [0, 1, 2, 4, 9, 11])
def test_fused_stem_pairs_run_time_count_variant(case, fuse_whatever_fits, monkeypatch):
    """Shapes without a static instantiation run the variant whose chunk / item counts are
    run-time values (every wait drains the queue, stores issued at once); forced here
    (CTG_STEM_GENERIC, read at every launch) on shapes that normally take a static one: same
    tables, same arithmetic per element -- the same bits as the static variant in the same
    (X / Y) form of step 2.  The row-interleaved form of round 4 (the default where it applies)
    adds the two products of an imaginary part in the other order: the last bit may differ,
    the gate against the oracle is the same."""
    nq, gates = G.STEM_CASES[case]
    tree = G.stem_network(nq, gates, 100 * case)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=case, dtype="complex64")
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    gate = G.single_gate(ref, orc.contract(tree, arrays))
    monkeypatch.setenv("CTG_STEM_BF16X3", "0")     # (fp32 products: what the run-time-count variant has)
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    default = np.asarray(fn(*arrays))
    names = [n for n in fn.setup(*arrays)["exec"].step_kernels() if n.startswith("stem2_kernel")]
    assert names
    monkeypatch.setenv("CTG_STEM_NO_RI2", "1")
    static = np.asarray(fn(*arrays))
    xnames = [n for n in fn.setup(*arrays)["exec"].step_kernels() if n.startswith("stem2_kernel")]
    assert all(not (f["bf3"] or f["ri2"] or f["one"]) for f in map(G.stem_flags, xnames)), xnames   # (fp32, X / Y form, a pair)
    monkeypatch.setenv("CTG_STEM_GENERIC", "1")
    generic = np.asarray(fn(*arrays))
    gnames = [n for n in fn.setup(*arrays)["exec"].step_kernels() if n.startswith("stem2_kernel")]
    assert all(G.stem_flags(n)["nch"] == 0 for n in gnames), gnames
    fn.close()
    assert np.array_equal(static, generic)
    assert G.relerr(default, ref) <= gate and G.relerr(static, ref) <= gate
    if any(G.stem_flags(n)["ri2"] for n in names):   # (the eleventh template argument: RI2)
        assert G.relerr(default, static) <= 4e-6



# ---------------------------------------------------------------------- #
# host robustness: threads, checkpoints, progress, cache bounds
# ---------------------------------------------------------------------- #


def test_threads_share_cached_expressions():
    """Two threads hammering ``ca.einsum`` with the same (cached) expression and different
    operands: every result is its own thread's (the contractor serialises upload -> run ->
    fetch; SURVEY 8b: handles thread-confined *or locked*)."""
    import threading

    from cotengra_amd import interface

    interface.clear_expression_cache()
    rng = np.random.default_rng(0)
    eq = "abc,cd,bde->ae"
    shapes = [(6, 5, 4), (4, 7), (5, 7, 3)]
    jobs = []
    for t in range(4):
        xs = [(rng.normal(size=s) + 1j * rng.normal(size=s)).astype("complex128") for s in shapes]
        jobs.append((xs, np.einsum(eq, *xs)))
    errors = []

    def work(t):
        xs, ref = jobs[t]
        try:
            for _ in range(40):
                got = np.asarray(ca.einsum(eq, *xs))
                if not np.allclose(got, ref, rtol=1e-10, atol=1e-12):
                    errors.append((t, float(np.abs(got - ref).max())))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert len(interface._EXPR_CACHE) == 1   # one expression served all of them


def test_expression_cache_is_bounded_by_device_bytes(monkeypatch):
    from cotengra_amd import interface

    interface.clear_expression_cache()
    rng = np.random.default_rng(1)
    # the first expression tells what one of these executors really holds (ctg_exec_device_bytes:
    # inputs, arena, tables, result -- a few hundred KB; no scratch: no step of theirs needs any);
    # the bound is set to three and a half of them
    a, b = rng.normal(size=(4, 64)), rng.normal(size=(64, 4))
    assert np.allclose(np.asarray(ca.einsum("ab,bc->ac", a, b)), a @ b)
    one = next(iter(interface._EXPR_CACHE.values())).device_bytes()
    assert 0 < one < (8 << 20)
    bound = int(3.5 * one)
    monkeypatch.setattr(interface, "_EXPR_CACHE_BYTES", bound)
    for n in range(5, 12):   # seven more shapes of about the same size
        a, b = rng.normal(size=(n, 64)), rng.normal(size=(64, n))
        assert np.allclose(np.asarray(ca.einsum("ab,bc->ac", a, b)), a @ b)
        assert sum(e.device_bytes() for e in interface._EXPR_CACHE.values()) <= bound + one or len(interface._EXPR_CACHE) == 1
    assert 1 <= len(interface._EXPR_CACHE) < 8
    # the executor's out-of-memory path drops everything but the expression in the making
    keep = next(reversed(interface._EXPR_CACHE.values()))
    interface.evict_expression_cache(keep=keep.fn)
    assert list(interface._EXPR_CACHE.values()) == [keep]
    interface.clear_expression_cache()


def test_checkpoint_refuses_other_inputs(tmp_path):
    case = next(c for c in G.cases("tree") if c["name"] == "lattice8x8_sliced")
    tree = G.tree_of(case)
    arrays = G.arrays_of(case, "complex128", tree)
    ck = str(tmp_path / "run.npz")
    assert tree.contract_resumable(arrays, ck, every=1, stop_after=2) is None
    other = [a.copy() for a in arrays]
    other[3] = other[3] * 2.0   # "another bitstring": same tree, different tensors
    with pytest.raises(ValueError, match="signature"):
        tree.contract_resumable(other, ck, every=1)
    seen = []
    out = tree.contract_resumable(arrays, ck, every=1, progbar=lambda d, n: seen.append((d, n)))
    assert G.relerr(out, G.expected("lattice8x8_sliced/complex128")) < 1e-10
    assert seen[0][0] == 2 and seen[-1] == (tree.nslices, tree.nslices)   # resumed at 2, counted to the end


def test_progress_counts_finished_slices():
    case = next(c for c in G.cases("tree") if c["name"] == "lattice8x8_sliced")
    tree = G.tree_of(case)
    arrays = G.arrays_of(case, "complex128", tree)
    seen = []
    out = tree.contract(arrays, progbar=lambda d, n: seen.append((d, n)))
    assert G.relerr(out, G.expected("lattice8x8_sliced/complex128")) < 1e-10
    assert seen and seen[-1] == (tree.nslices, tree.nslices)
    assert all(b[0] > a[0] for a, b in zip(seen, seen[1:]))
    quiet = tree.contract(arrays)   # same bits with and without the counter
    assert np.array_equal(np.asarray(out), np.asarray(quiet))


def test_plain_c_driver_of_the_collective(tmp_path):
    """tests/cabi_reduce.c: plan -> exec -> upload -> unique id -> comm -> run_slices(first =
    rank, stride = world) -> reduce -> download, from plain C with no Python in the process;
    alone (world = 1: every collective still runs) and with the id handed through a file."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cabi_reduce")
    if not os.path.exists(exe):
        import __graft_entry__ as g

        g.build()
    plan = os.path.join(root, "tests", "golden", "cabi_plan.bin")
    for extra in ([], ["0", "1", str(tmp_path / "id"), "0"]):
        r = subprocess.run([exe, plan] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout, r.stderr)
        assert "rel err" in r.stdout and "4 of 4 slices" in r.stdout


# ---------------------------------------------------------------------- #
# float32 / float64 matrix-core kernel: 16-byte gathers where the layout allows
# ---------------------------------------------------------------------- #

REAL_CASES = [
    ("ab,bc->ac", dict(a=256, b=128, c=192)),     # A along k, B along its columns: both in pieces
    ("ab,cb->ac", dict(a=256, b=128, c=192)),     # B along k
    ("ba,bc->ac", dict(a=256, b=128, c=192)),     # A along its rows
    ("ba,cb->ca", dict(a=192, b=256, c=128)),     # rows of A fastest, k of B fastest, output transposed
    ("ab,bc->ac", dict(a=130, b=66, c=70)),       # extents that are multiples of 2 only: double yes, float no
    ("ab,bc->ac", dict(a=129, b=65, c=67)),       # odd extents: element-wise path
    ("xab,xbc->xac", dict(x=3, a=128, b=64, c=64)),   # batch index
    ("abk,kcd->abcd", dict(a=32, b=16, k=64, c=8, d=16)),   # fused index groups
]


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("case", range(len(REAL_CASES)))
def test_real_kernel_vector_gathers(case, dtype):
    from cotengra_amd.interface import einsum

    eq, sizes = REAL_CASES[case]
    (ta, tb), out = ca.eq_to_inputs_output(eq)
    rng = np.random.default_rng(case)
    a, b = (rng.normal(size=[sizes[i] for i in t]).astype(dtype) for t in (ta, tb))
    ref = np.einsum(eq, a.astype("float64"), b.astype("float64"), optimize=True)
    got = np.asarray(einsum(eq, a, b, optimize=[(0, 1)]))
    assert got.shape == ref.shape and got.dtype == np.dtype(dtype)
    tol = 1e-12 if dtype == "float64" else G.single_gate(ref, np.einsum(eq, a, b, optimize=True))
    assert G.relerr(got, ref) <= tol, (G.relerr(got, ref), tol)
    # a sliced operand view (odd base offset for some slices): falls back or stays correct
    sl = next(ix for ix in ta if ix in tb)
    tree = ca.ContractionTree.from_path([ta, tb], out, sizes, path=[(0, 1)])
    tree.remove_ind_(sl)
    assert G.relerr(np.asarray(tree.contract([a, b])), ref) <= max(tol, 1e-12) * 4
