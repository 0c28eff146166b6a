"""Round 6 on the GPU: LDS-resident subtrees (csrc/ctg_lds_run.hip) -- one workgroup walks a whole subtree
whose tensors fit a compute unit's LDS.  Against the oracle in four dtypes, against the step-by-step path
(bit for bit where the arithmetic order is the same), under slice batching, slice groups and
strip_exponent (which falls back to the ordinary steps)."""
import os

import numpy as np
import pytest

import cotengra_amd as ca
from cotengra_amd import plan as P
from cotengra_amd.contractor import HipContractor
from oracle import contract_ref as orc

import golden_util as G

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = {"float64": 1e-10, "complex128": 1e-10}


def case_of(name):
    return next(c for c in G.cases("tree") if c["name"] == name)


def contract(tree, arrays, **kw):
    fn = HipContractor(tree)
    try:
        out = fn(*arrays, **kw)
        st = fn.setup(*arrays)
        ex, plan = st["exec"], st["plan"]
        info = {"launches": ex.launch_count(), "kernels": ex.step_kernels(), "plan": plan, "batch": ex.batch}
    finally:
        fn.close()
    if isinstance(out, tuple):
        return (np.asarray(out[0]), out[1]), info
    return np.asarray(out), info


LDS_TREES = ["C1_rand10_d4", "C2_lattice8x8_d4", "lattice8x8_sliced", "lattice4x4_sliced", "preproc_s0_a", "preproc_s1",
             "rand_s42_r3_o2_hi1_ho2", "rand_s666_r3_o2_hi2_ho2_sliced", "rand_s42_r2_o2_hi0_ho2_outsliced",
             "project_1", "C5_hyper200"]


@pytest.mark.parametrize("name", LDS_TREES)
@pytest.mark.parametrize("dtype", ["complex64", "complex128", "float32", "float64"])
def test_subtrees_in_lds_against_oracle_and_step_path(name, dtype, monkeypatch):
    names = {c["name"] for c in G.cases("tree")}
    if name not in names:
        pytest.skip("no such golden tree")
    case = case_of(name)
    tree = G.tree_of(case)
    wide = "complex128" if dtype.startswith("complex") else "float64"
    if wide == "float64" and "float64" not in case["dtypes"]:
        pytest.skip("complex-only case")
    arrays = G.arrays_of(case, wide, tree)
    big = tree.nslices > 64
    ids = list(range(0, 6)) if big else None

    def run(env):
        for k in ("CTG_NO_LDS_RUNS", "CTG_LDS_NO_MFMA"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        xs = [a.astype(dtype) for a in arrays]
        if ids is None:
            return contract(tree, xs)
        fn = HipContractor(tree)
        try:
            st = fn.setup(*xs)
            ex = st["exec"]
            ex.zero_result()
            ex.run_slice_list(ids)
            out = np.asarray(ex.download_result())
            info = {"launches": ex.launch_count(), "kernels": ex.step_kernels(), "plan": st["plan"], "batch": ex.batch}
        finally:
            fn.close()
        return out, info

    got, info = run({})
    plan = info["plan"]
    if not plan.lds_runs:
        pytest.skip("the planner found no LDS-resident subtree here")
    members = [i for i, s in enumerate(plan.steps) if s.lds_comp >= 0]
    assert all(info["kernels"][i].startswith("lds_run_kernel") for i in members), info["kernels"]
    old, info_old = run({"CTG_NO_LDS_RUNS": "1"})
    assert not any(k.startswith("lds_run_kernel") for k in info_old["kernels"])
    if name in ("C1_rand10_d4", "C2_lattice8x8_d4", "C5_hyper200", "lattice8x8_sliced"):
        # (a tree of ten tensors can lose a wave-front group to the split and gain a launch; the trees the
        # model is for must need fewer)
        assert info["launches"][1] < info_old["launches"][1], (info["launches"], info_old["launches"])
    if ids is None:
        ref = np.asarray(orc.contract(tree, arrays))
        if dtype in TOL:
            tol = TOL[dtype]
        else:
            tol = G.single_gate(ref, orc.contract(tree, [a.astype(dtype) for a in arrays]))
        if np.abs(ref).max() > 1e-30 or dtype in TOL:
            assert G.relerr(got.reshape(np.shape(ref)), ref) <= tol, (G.relerr(got.reshape(np.shape(ref)), ref), tol)
            assert G.relerr(old.reshape(np.shape(ref)), ref) <= tol
    else:
        # (a few slices of a tree with 4e9 of them, output-sliced: the partial result of the step-by-step path
        # is the reference here; the slices themselves are pinned to the oracle in test_gpu_golden.py)
        assert G.relerr(got, old) <= (1e-10 if dtype in TOL else 1e-5)
    # the same bits as the ordinary steps where those run on the thread-per-output kernel (same sums, same order)
    same_order = all(info_old["kernels"][i] in ("pair_valu_kernel", "single_kernel") and plan.steps[i].K < 256 for i in members)
    rest_same = True   # (the steps outside the components are the same launches either way)
    if same_order and rest_same and info["batch"] == 1:
        # (complex64 steps with >= 8 columns run on the matrix cores inside the component: switched off here)
        plain, _ = run({"CTG_LDS_NO_MFMA": "1"}) if dtype == "complex64" else (got, None)
        assert np.array_equal(plain, old), name


def test_c2_launches_and_result():
    """BASELINE config C2: the 8 x 8 lattice.  48 of its 63 steps run in LDS, the launch count drops."""
    case = case_of("C2_lattice8x8_d4")
    tree = G.tree_of(case)
    arrays = G.arrays_of(case, "complex128", tree)
    ref = complex(np.asarray(orc.contract(tree, arrays)))
    xs = [a.astype("complex64") for a in arrays]
    got, info = contract(tree, xs)
    steps, launches = info["launches"]
    assert steps == len(info["plan"].steps) and launches <= 16, info["launches"]
    assert abs(complex(got) - ref) <= G.single_gate(ref, orc.contract(tree, xs)) * abs(ref)


def test_m10_amplitude_batched_slices_through_lds():
    """BASELINE config C3: 64 slices of the Sycamore m10 amplitude in one batch of launches, the per-slice
    subtrees in LDS (blockIdx.y = slice)."""
    tree = ca.tree_from_record(ca.load_network(os.path.join(ROOT, "tests/golden/trees/sycamore_m10.json")))
    z = np.load(os.path.join(ROOT, "tests/golden/sycamore_m10_arrays.npz"))
    arrays = [z[f"t{i}"] for i in range(tree.N)]
    ex = np.load(os.path.join(ROOT, "tests/golden/sycamore_m10_expected.npz"))
    ref = complex(ex["amplitude"])
    xs = [a.astype("complex64") for a in arrays]
    got, info = contract(tree, xs)
    assert info["plan"].lds_runs and info["batch"] > 1
    assert any(k.startswith("lds_run_kernel") for k in info["kernels"])
    assert abs(complex(got) - ref) <= max(1e-5, 8 * abs(complex(orc.contract(tree, xs)) - ref) / abs(ref)) * abs(ref)


def test_strip_exponent_runs_the_ordinary_steps():
    case = case_of("C2_lattice8x8_d4")
    tree = G.tree_of(case)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=7, dtype="complex128", rescale=False)
    m_ref, e_ref = orc.contract(tree, arrays, strip_exponent=True)
    (m, e), _ = contract(tree, [a.astype("complex64") for a in arrays], strip_exponent=True)
    lg = np.log10(abs(complex(m))) + e
    lg_ref = np.log10(abs(complex(m_ref))) + e_ref
    assert abs(lg - lg_ref) < 1e-4
    z, z_ref = complex(m) / abs(complex(m)), complex(m_ref) / abs(complex(m_ref))
    assert abs(z - z_ref) < 1e-3


# ---------------------------------------------------------------------- #
# the reference's own input scale at Sycamore depth (missing 2)
# ---------------------------------------------------------------------- #

TREES = os.path.join(ROOT, "tests", "golden", "trees")


def _strip_sum(pairs):
    """Sum of (mantissa, exponent) pairs as the reference's adder forms it (core.py:125-172), in Python
    complex: ``(mantissa, exponent)`` with the largest exponent."""
    emax = max(e for _, e in pairs)
    return sum(complex(m) * 10.0 ** (e - emax) for m, e in pairs), emax


@pytest.mark.parametrize("fixture", ["sycamore_m20_native.json", "sycamore_m10.json"])
@pytest.mark.parametrize("mode", ["complex64-bf16x3", "complex64-fp32", "complex128"])
def test_raw_inputs_under_strip_exponent_at_sycamore_depth(fixture, mode, monkeypatch):
    """The reference normalises after EVERY step under strip_exponent (contract.py:816-829), so its own
    un-rescaled inputs (Frobenius-normalised, utils.py:1243-1284) go through a 380-step m20 tree in
    complex64; here normalisation is lazy.  Trees narrowed to width 2^20, ``rescale=False``,
    ``strip_exponent=True``: 64 slices summed on the device and single slices, both arithmetics and
    double precision, against the oracle's (mantissa, exponent) in complex128 -- compared as
    log10|m| + e and as the normalised mantissa."""
    tree = ca.tree_from_record(ca.load_network(os.path.join(TREES, fixture)))
    small = tree.slice(target_size=2**20) if tree.max_size() > 2**20 else tree
    a128 = ca.make_arrays_from_inputs(small.inputs, small.size_dict, seed=42, dtype="complex128", rescale=False)
    dtype = "complex128" if mode == "complex128" else "complex64"
    if mode != "complex128":
        monkeypatch.setenv("CTG_STEM_BF16X3", "1" if mode.endswith("bf16x3") else "0")
    xs = [a.astype(dtype) for a in a128]
    n = int(min(small.nslices, 2**40))
    ids = sorted({int(i) for i in np.linspace(0, n - 1, 64)})
    assert len(ids) >= min(64, n)
    ref_pairs = [orc.contract_slice(small, a128, i, strip_exponent=True) for i in ids]
    s_ref, e_ref = _strip_sum(ref_pairs)
    assert abs(s_ref) > 0 and e_ref < -30, "the value must lie far below the float32 range for this test to bite"
    if dtype == "complex64":
        s_np, e_np = _strip_sum([orc.contract_slice(small, xs, i, strip_exponent=True) for i in ids])
        np_err = abs(s_np * 10.0 ** (e_np - e_ref) - s_ref) / abs(s_ref)
        gate = max(1e-5, 8.0 * np_err)
    else:
        gate = 1e-10

    def check(m, e, s, es, tol):
        lg, lg_ref = np.log10(abs(complex(m))) + e, np.log10(abs(s)) + es
        assert abs(lg - lg_ref) <= max(tol, 1e-12) / np.log(10.0) * 1.5 + 1e-12, (lg, lg_ref)
        z, z_ref = complex(m) / abs(complex(m)), s / abs(s)
        assert abs(z - z_ref) <= 1.5 * tol, (abs(z - z_ref), tol)

    fn = HipContractor(small)
    try:
        st = fn.setup(*xs)
        ex = st["exec"]
        ex.set_strip_exponent(True, False)
        ex.zero_result()
        ex.run_slice_list(ids)
        part, e, zero = ex.get_state()
        assert not zero
        check(part, e, s_ref, e_ref, gate)
        # single slices
        for k in (0, len(ids) - 1):
            m1, e1 = fn.contract_slice(xs, ids[k], strip_exponent=True)
            sm, se = complex(ref_pairs[k][0]), ref_pairs[k][1]
            tol1 = gate
            if dtype == "complex64":
                mn, en = orc.contract_slice(small, xs, ids[k], strip_exponent=True)
                tol1 = max(1e-5, 8.0 * abs(complex(mn) * 10.0 ** (en - se) - sm) / abs(sm))
            check(np.asarray(m1), e1, sm, se, tol1)
    finally:
        fn.close()


# ---------------------------------------------------------------------- #
# multi-GPU path hardened for the day an 8-GPU node exists
# ---------------------------------------------------------------------- #


def test_bench_line_is_complete_when_rccl_cannot_be_loaded():
    """``bench.py`` under a launcher with ``CTG_RCCL_LIB`` pointing at a file that does not exist:
    ``ctg_comm_init`` fails loudly (CTG_E_COMM), every rank agrees to reduce through torch.distributed instead,
    and the line is still complete -- contract keys, per-rank spread, share sizes, the N > 1 legs."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, CTG_BENCH_C3_AMPLITUDES="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               CTG_RCCL_LIB="/nonexistent/librccl-missing.so")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29519", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "ctg_comm_init failed" in r.stderr
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 1 and rec["roofline"]["frac"] > 0 and rec["value"] > 0
    assert rec["config"]["reduce_via"] == "torch.distributed.reduce (RCCL)"
    assert rec["ranks"]["share_units"] == [2**20 // rec["ranks"]["slices_per_unit"]]
    assert len(rec["ranks"]["slices_ms"]) == 1 and rec["ranks"]["reduce_wait_ms"][0] >= 0
    assert rec["legs"]["C3_strong_ms"] > 0 and rec["legs"]["C3_amplitudes_per_sec"] > 50


def test_plain_c_driver_fails_loudly_without_rccl(tmp_path):
    """``tests/cabi_reduce`` (no Python) with RCCL unloadable: a non-zero exit and the library's message, not a
    silent single-GPU result."""
    import subprocess

    exe = os.path.join(ROOT, "tests", "cabi_reduce")
    plan = os.path.join(ROOT, "tests", "golden", "cabi_plan.bin")
    if not os.path.exists(exe):
        pytest.skip("tests/cabi_reduce not built")
    env = dict(os.environ, CTG_RCCL_LIB="/nonexistent/librccl-missing.so")
    r = subprocess.run([exe, plan], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "rccl" in (r.stdout + r.stderr).lower()


# ---------------------------------------------------------------------- #
# the fp16 x 2 arithmetic of the stem kernels (csrc/ctg_stem.hip built with -DCTG_STEM_H2): two rounded fp16 limbs
# per operand under per-tensor power-of-two scales, three products -- what a new executor multiplies its pairs with
# ---------------------------------------------------------------------- #


@pytest.fixture
def fuse_whatever_fits(monkeypatch):
    from cotengra_amd import stem
    monkeypatch.setattr(stem, "gather_rate", lambda run_bytes: 5.4e12)
    for k in ("CTG_STEM_ARITH", "CTG_STEM_BF16X3", "CTG_STEM_H2"):
        monkeypatch.delenv(k, raising=False)
    # (by default the FIRST pair of a stem -- its big operand comes from a kernel that records no maximum -- is
    # multiplied in bf16 x 3; the kernel tests want every capable pair in fp16 x 2: a max-abs pass supplies the scale)
    monkeypatch.setenv("CTG_STEM_H2_ALL", "1")


def _stem_names(fn, arrays):
    return [n for n in fn.setup(*arrays)["exec"].step_kernels() if n.startswith(("stem2_kernel", "stem2h_kernel"))]


@pytest.mark.parametrize("sliced", [0, 2])
@pytest.mark.parametrize("case", range(len(G.STEM_CASES)))
def test_stem_pairs_in_fp16x2(case, sliced, fuse_whatever_fits, monkeypatch):
    """Every instantiation of the fused kernel: the default arithmetic is fp16 x 2 wherever the shape has a
    16-bit-pipe kernel (names stem2h_kernel<..., BF3 = true ...>), within the single-precision gate of the oracle;
    "bf16x3" / "fp32" by option give the other arithmetics' bits, CTG_STEM_H2=0 the bf16 x 3 bits; strip_exponent
    falls back to bf16 x 3 (a scale per step there)."""
    nq, gates = G.STEM_CASES[case]
    tree = G.stem_network(nq, gates, 100 * case, sliced=sliced)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=case, dtype="complex64")
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    gate = G.single_gate(ref, orc.contract(tree, arrays))
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    got = np.asarray(fn(*arrays))
    names = _stem_names(fn, arrays)
    assert names
    h2 = [n for n in names if n.startswith("stem2h_kernel")]
    assert all(G.stem_flags(n)["bf3"] for n in h2)
    assert G.relerr(got, ref) <= gate, (G.relerr(got, ref), gate)
    fb = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10, stem_bf16x3="bf16x3")
    bf3 = np.asarray(fb(*arrays))
    names_b = _stem_names(fb, arrays)
    fb.close()
    assert not any(n.startswith("stem2h_kernel") for n in names_b)
    # (the same instantiations -- but for the form of a pair: specialised waves under two limbs, where they win;
    # the symmetric kernel under three, where they lose: the last template argument)
    def shape_of(n):
        return n[n.index("<") + 1: n.rindex(">")].split(",")[:16]
    assert [shape_of(n) for n in names] == [shape_of(n) for n in names_b]
    assert G.relerr(bf3, ref) <= gate
    if h2:
        assert not np.array_equal(got, bf3)   # (it really ran)
    monkeypatch.setenv("CTG_STEM_H2", "0")
    assert np.array_equal(np.asarray(fn(*arrays)), bf3)
    monkeypatch.delenv("CTG_STEM_H2")
    m, e = fn(*arrays, strip_exponent=True)
    assert not any(n.startswith("stem2h_kernel") for n in _stem_names(fn, arrays))
    assert G.relerr(np.asarray(m).astype("complex128") * 10.0**e, ref) <= gate
    assert np.array_equal(np.asarray(fn(*arrays)), got)     # (and back)
    if tree.nslices > 1:
        a128 = [a.astype("complex128") for a in arrays]
        for i in range(tree.nslices):
            ri = np.asarray(orc.contract_slice(tree, a128, i))
            gi = max(gate, G.single_gate(ri, orc.contract_slice(tree, arrays, i)))
            assert G.relerr(np.asarray(fn.contract_slice(arrays, i)), ri) <= gi
    fn.close()


@pytest.mark.parametrize("seed", range(48))
def test_random_stems_in_fp16x2(seed, fuse_whatever_fits):
    tree = G.random_stem(seed)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=seed, dtype="complex64")
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    gate = G.single_gate(ref, orc.contract(tree, arrays))
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 9)
    got = np.asarray(fn(*arrays))
    fn.close()
    assert G.relerr(got, ref) <= gate, (G.relerr(got, ref), gate)


@pytest.mark.parametrize("case", range(len(G.ONE_CASES)))
def test_single_stem_steps_in_fp16x2(case, fuse_whatever_fits, monkeypatch):
    # (the pairing model takes a single step only where it pays; here: wherever the kernel can)
    from cotengra_amd import stem
    monkeypatch.setattr(stem, "single_seconds", lambda *a, **k: 0.0)
    nq, gates = G.ONE_CASES[case]
    tree = G.stem_network(nq, gates, 300 + case)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=case, dtype="complex64")
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    gate = G.single_gate(ref, orc.contract(tree, arrays))
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    got = np.asarray(fn(*arrays))
    names = _stem_names(fn, arrays)
    fn.close()
    assert names
    assert G.relerr(got, ref) <= gate, (G.relerr(got, ref), gate)


def test_fp16x2_scales_are_exact_powers_of_two(fuse_whatever_fits):
    """Inputs alternately scaled by 2^+40 and 2^-40, the big state by 2^-90: every scale the kernels take out is a
    power of two found from the data, so the limbs -- and the result, up to the power put back in -- are the SAME
    BITS as in the plain run; a chain of pairs whose intermediates' magnitudes differ by 2^40 from one pair to the
    next also checks that every pair scales with ITS operand's record, not a stale one."""
    nq, gates = G.STEM_CASES[10]
    tree = G.stem_network(nq, gates, 1000)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=10, dtype="complex64", rescale=True)
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    plain = np.asarray(fn(*arrays))
    assert any(n.startswith("stem2h_kernel") for n in _stem_names(fn, arrays))
    shifts = [(-90 if i == 0 else (40 if i % 2 else -40)) for i in range(len(arrays))]
    scaled = [(a * np.float32(2.0**s)).astype("complex64") for a, s in zip(arrays, shifts)]
    got = np.asarray(fn(*scaled))
    fn.close()
    total = sum(shifts)
    assert np.array_equal(got * np.float32(2.0 ** -total) if abs(total) < 120 else got, plain) or \
        G.relerr(got.astype("complex128") * 2.0 ** -total, ref) <= 1e-5
    assert G.relerr(got.astype("complex128") * 2.0 ** -total, ref) <= G.single_gate(ref, orc.contract(tree, arrays))


def test_fp16x2_wide_dynamic_range_is_norm_wise(fuse_whatever_fits):
    """fp16 limbs cover 2^-24 ... 2^16 of a tensor's largest element: an operand whose elements span 2^120 keeps its
    large ones to 22 bits and loses the small ones gradually.  What holds is the NORM-WISE bound (error relative to the
    largest element of the result) -- the exact three-way bf16 split keeps every element to fp32 accuracy and is the
    arithmetic for such data (stem_bf16x3="bf16x3", CTG_STEM_ARITH)."""
    nq, gates = G.STEM_CASES[4]
    tree = G.stem_network(nq, gates, 404)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=4, dtype="complex64")
    rng = np.random.default_rng(7)
    big = arrays[0]
    arrays[0] = (big * np.exp2(rng.integers(-60, 61, size=big.shape)).astype("float32")).astype("complex64")
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    got = np.asarray(fn(*arrays)).astype("complex128")
    assert any(n.startswith("stem2h_kernel") for n in _stem_names(fn, arrays))
    fn.close()
    fb = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10, stem_bf16x3="bf16x3")
    bf3 = np.asarray(fb(*arrays)).astype("complex128")
    fb.close()
    top = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 1e-5 * top          # norm-wise: fp16 x 2
    assert np.abs(bf3 - ref).max() <= 1e-5 * top
    # (on THIS data the outputs are sums dominated by their large terms and the two arithmetics are equally good
    # element by element -- median relative error 2e-7 either way; the difference is in what is promised)


def test_first_pair_of_a_stem_runs_bf16x3_and_records_its_maximum(monkeypatch):
    """The default rule: fp16 x 2 needs the largest element of the big operand, which only a stem launch of the 16-bit
    pipe records -- the first pair of a chain (operand of other origin) runs bf16 x 3 and records, every later pair runs
    fp16 x 2 on its producer's record.  Against the oracle; and with CTG_STEM_H2_ALL=1 (a max-abs pass for the first
    pair) the result changes only in the last bits."""
    from cotengra_amd import stem
    monkeypatch.setattr(stem, "gather_rate", lambda run_bytes: 5.4e12)
    for k in ("CTG_STEM_ARITH", "CTG_STEM_BF16X3", "CTG_STEM_H2", "CTG_STEM_H2_ALL"):
        monkeypatch.delenv(k, raising=False)
    nq, gates = G.STEM_CASES[10]     # a longer stem: pairs + leftovers
    tree = G.stem_network(nq, gates, 1000)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=10, dtype="complex64")
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    gate = G.single_gate(ref, orc.contract(tree, arrays))
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    got = np.asarray(fn(*arrays))
    names = _stem_names(fn, arrays)
    assert names[0].startswith("stem2_kernel<") and any(n.startswith("stem2h_kernel<") for n in names[1:]), names
    assert G.relerr(got, ref) <= gate
    monkeypatch.setenv("CTG_STEM_H2_ALL", "1")
    allh = np.asarray(fn(*arrays))
    assert _stem_names(fn, arrays)[0].startswith("stem2h_kernel<")
    fn.close()
    assert G.relerr(allh, ref) <= gate and G.relerr(allh, got) <= 1e-5


# ---------------------------------------------------------------------- #
# long tiled steps in the fp16 x 2 arithmetic (csrc/ctg_pair_mfma.hip: pair_mfma_h2_kernel)
# ---------------------------------------------------------------------- #


def _chain_tree(R, K, N, N2=None):
    """a[R, K] b[K, N] (c[N, N2]): one long tiled step, or two of them in a chain (the second one's big operand
    produced -- and its largest element recorded -- by the first)."""
    if N2 is None:
        return ca.ContractionTree.from_path([("a", "b"), ("b", "c")], ("a", "c"), dict(a=R, b=K, c=N), path=[(0, 1)])
    return ca.ContractionTree.from_path([("a", "b"), ("b", "c"), ("c", "d")], ("a", "d"), dict(a=R, b=K, c=N, d=N2),
                                        path=[(0, 1), (0, 1)])


def _cplx(rng, *shape):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype("complex64")


@pytest.mark.parametrize("R,K,N", [(8192, 512, 512), (65536, 64, 64), (32768, 256, 128)])
def test_long_tiled_steps_in_fp16x2(R, K, N, monkeypatch):
    """A new executor's arithmetic is fp16 x 2: a GEMM-like complex64 step with K >= 64 on full 64-column tiles runs
    pair_mfma_h2_kernel -- both operands split into two rounded fp16 limbs under a power of two per TENSOR (a max-abs
    pass here: the operands are inputs), three products.  Against the complex128 oracle, error relative to the size of
    the terms: random data and a contraction that cancels by 2^-12 within 3 x the bf16 x 3 kernel's error (same data,
    CTG_PAIR_H2=0); operands scaled by 2^+40 / 2^-37 give the same bits up to the power put back in."""
    for k in ("CTG_STEM_ARITH", "CTG_STEM_BF16X3", "CTG_PAIR_BF16X3", "CTG_PAIR_H2"):
        monkeypatch.delenv(k, raising=False)
    tree = _chain_tree(R, K, N)
    rng = np.random.default_rng(R + K + N)
    a, b = _cplx(rng, R, K), _cplx(rng, K, N)
    a_canc = a.copy()
    a_canc[:, K // 2:] = -a[:, : K // 2]
    b_canc = b.copy()
    b_canc[K // 2:, :] = b[: K // 2, :] * np.float32(1.0 + 2.0**-12)
    fn = HipContractor(tree)
    for label, (x, y) in {"random": (a, b), "cancelling": (a_canc, b_canc)}.items():
        ref = x.astype("complex128") @ y.astype("complex128")
        terms = np.abs(x.astype("complex128")) @ np.abs(y.astype("complex128"))
        got = np.asarray(fn(x, y))
        names = fn.setup(x, y)["exec"].step_kernels()
        assert any(n.startswith("pair_mfma_h2_kernel") for n in names), names
        monkeypatch.setenv("CTG_PAIR_H2", "0")
        bf3 = np.asarray(fn(x, y))
        assert any(n.startswith("pair_mfma_bf3_kernel") for n in fn.setup(x, y)["exec"].step_kernels())
        monkeypatch.delenv("CTG_PAIR_H2")
        e_h2 = float((np.abs(got - ref) / terms).max())
        e_b3 = float((np.abs(bf3 - ref) / terms).max())
        print(label, f"fp16x2 {e_h2:.2e} bf16x3 {e_b3:.2e}")
        assert not np.array_equal(got, bf3)          # (it really ran)
        assert e_h2 <= max(3.0 * e_b3, 5e-7), (label, e_h2, e_b3)
    plain = np.asarray(fn(a, b))
    scaled = np.asarray(fn((a * np.float32(2.0**40)).astype("complex64"), (b * np.float32(2.0**-37)).astype("complex64")))
    fn.close()
    assert np.array_equal(scaled * np.float32(2.0**-3), plain)


def test_chained_tiled_steps_take_the_producers_record(monkeypatch):
    """Two long tiled steps in a chain: the first one records the largest element it stores, the second one splits
    its big operand under that record (no max-abs pass over it) -- with the first result 2^50 away from the inputs'
    magnitude, a stale or missing record would overflow or flush the limbs.  Against the oracle, norm-wise."""
    for k in ("CTG_STEM_ARITH", "CTG_STEM_BF16X3", "CTG_PAIR_BF16X3", "CTG_PAIR_H2"):
        monkeypatch.delenv(k, raising=False)
    R, K, N, N2 = 16384, 256, 256, 256
    tree = _chain_tree(R, K, N, N2)
    rng = np.random.default_rng(5)
    a, b, c = _cplx(rng, R, K), _cplx(rng, K, N), _cplx(rng, N, N2)
    a = (a * np.float32(2.0**25)).astype("complex64")
    b = (b * np.float32(2.0**25)).astype("complex64")
    c = (c * np.float32(2.0**-45)).astype("complex64")
    ref = (a.astype("complex128") @ b.astype("complex128")) @ c.astype("complex128")
    fn = HipContractor(tree)
    got = np.asarray(fn(a, b, c))
    names = fn.setup(a, b, c)["exec"].step_kernels()
    fn.close()
    assert sum(n.startswith("pair_mfma_h2_kernel") for n in names) == 2, names
    assert G.relerr(got, ref) <= 2e-6, G.relerr(got, ref)


def test_batched_tiled_steps_scale_every_slice_by_its_own_record(monkeypatch):
    """Four slices of (a0[s, R, j] m[j, K]) b[K, N] w[s] in ONE launch sequence, slice s larger than slice s - 1 by
    2^30: the long tiled step of the batch splits every slice's intermediate under THAT slice's largest element (a
    record per step and slice of the batch) -- one record for the whole batch would leave the small slices' limbs in
    fp16's subnormals.  Every slice against the oracle; and the same bits whatever the batch size.  (An INPUT read in
    place by a tiled step is scaled by the largest element of the whole leaf: the second tree.)"""
    for k in ("CTG_STEM_ARITH", "CTG_STEM_BF16X3", "CTG_PAIR_BF16X3", "CTG_PAIR_H2"):
        monkeypatch.delenv(k, raising=False)
    S, R, J, K, N = 4, 4096, 4, 256, 256
    rng = np.random.default_rng(11)
    a0 = _cplx(rng, S, R, J)
    for s in range(S):
        a0[s] *= np.float32(2.0 ** (30 * s - 60))
    m, b = _cplx(rng, J, K), _cplx(rng, K, N)
    w = np.ones(S, dtype="complex64")
    outs = {}
    for cap in ("4", "1"):
        monkeypatch.setenv("CTG_SLICE_BATCH", cap)
        tree = ca.ContractionTree.from_path([("s", "a", "j"), ("j", "b"), ("b", "c"), ("s",)], ("a", "c"),
                                            dict(s=S, a=R, j=J, b=K, c=N), path=[(0, 1), (0, 2), (0, 1)])
        tree.remove_ind_("s")
        assert tree.nslices == S
        fn = HipContractor(tree)
        ex = fn.setup(a0, m, b, w)["exec"]
        assert any(n.startswith("pair_mfma_h2_kernel") for n in ex.step_kernels()), ex.step_kernels()
        per_slice = []
        for i in range(S):
            ex.zero_result()
            ex.run_slices(i, 1, 1)
            per_slice.append(np.asarray(ex.download_result()).copy())
        ex.zero_result()
        ex.run_slices(0, S, 1)
        outs[cap] = (per_slice, np.asarray(ex.download_result()).copy())
        fn.close()
    for i in range(S):
        ref = (a0[i].astype("complex128") @ m.astype("complex128")) @ b.astype("complex128")
        assert G.relerr(outs["4"][0][i], ref) <= 2e-6, (i, G.relerr(outs["4"][0][i], ref))
        assert np.array_equal(outs["4"][0][i], outs["1"][0][i])
    assert np.array_equal(outs["4"][1], outs["1"][1])
    # a sliced input as the tiled step's own operand
    monkeypatch.setenv("CTG_SLICE_BATCH", "4")
    a = _cplx(rng, S, R, K)
    tree = ca.ContractionTree.from_path([("s", "a", "b"), ("b", "c"), ("s",)], ("a", "c"), dict(s=S, a=R, b=K, c=N),
                                        path=[(0, 1), (0, 1)])
    tree.remove_ind_("s")
    fn = HipContractor(tree)
    ex = fn.setup(a, b, w)["exec"]
    assert any(n.startswith("pair_mfma_h2_kernel") for n in ex.step_kernels())
    for ids in ((1, 1), (0, S)):
        ex.zero_result()
        ex.run_slices(ids[0], ids[1], 1)
        got = np.asarray(ex.download_result())
        ref = sum(a[i].astype("complex128") @ b.astype("complex128") for i in range(ids[0], ids[0] + ids[1]))
        assert G.relerr(got, ref) <= 2e-6
    fn.close()


@pytest.mark.parametrize("mode", ["side_by_side", "one_after_the_other"])
def test_arena_placement_does_not_change_a_bit(mode, monkeypatch, capfd):
    """A big single-slice arena is allocated twice (or, where it does not fit twice, one allocation after the other) and
    the copy that reads the plan's largest tensors faster is kept (ctg_runtime.hip: place_arena).  With the size threshold
    lowered to a small tree: the probe runs, an arena comes out of it, and the result is the same bits as without it."""
    rec = ca.load_network(os.path.join(ROOT, "tests", "golden", "trees", "sycamore_m10.json"))
    z = np.load(os.path.join(ROOT, "tests", "golden", "sycamore_m10_arrays.npz"))
    monkeypatch.setenv("CTG_SLICE_BATCH", "1")
    outs = {}
    for place in ("0", "1"):
        monkeypatch.setenv("CTG_ARENA_PLACE", place)
        monkeypatch.setenv("CTG_ARENA_PLACE_MIN", "1024")
        monkeypatch.setenv("CTG_ARENA_DEBUG", "1")
        if mode == "one_after_the_other":
            monkeypatch.setenv("CTG_ARENA_PLACE_SEQ", "1")
        tree = ca.tree_from_record(rec)
        xs = [z[f"t{i}"].astype("complex64") for i in range(tree.N)]
        fn = HipContractor(tree)
        outs[place] = complex(np.asarray(fn(*xs)))
        fn.close()
        err = capfd.readouterr().err
        assert ("arena placement" in err) == (place == "1"), err
        if place == "1":
            assert ("side by side" if mode == "side_by_side" else "one after the other") in err and "probe failed" not in err, err
    assert outs["0"] == outs["1"]
    ref = complex(np.load(os.path.join(ROOT, "tests", "golden", "sycamore_m10_expected.npz"))["amplitude"])
    assert abs(outs["1"] - ref) <= 1e-5 * abs(ref)
