"""GPU suite, round 4.

* ``sort_contraction_indices`` through the expression front end and the tree on the HIP
  path: reference-frozen values (tests/golden/gen/make_golden_r4.py);
* the row-interleaved step 2 of the fused stem kernel (csrc/ctg_stem.hip: RI2) really
  runs on the shapes it is instantiated for, and an INTERMEDIATE tensor of a pair --
  exposed by an identity second operand -- is checked element-wise against complex128;
* bf16 x 3 (fp32 products as six bf16 products): where it could fail -- 2^+-60 of dynamic
  range inside one operand, hard cancellation, whole operands near the bottom of the fp32
  range (third limbs below bf16's subnormals), un-rescaled inputs under ``strip_exponent``
  -- element-wise against the fp32 kernel and the complex128 oracle (max and RMS);
* two GPUs when the box has them: the plain-C rank driver and ``bench.py --gpus 2``.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import cotengra_amd as ca
from cotengra_amd.contractor import HipContractor
from cotengra_amd.plan import KIND_STEM2
from oracle import contract_ref as orc

import golden_util as G
from test_host_round4 import EXPECTED, SORTED, build

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------- #
# sort_contraction_indices on the device
# ---------------------------------------------------------------------- #


@pytest.mark.parametrize("name", sorted(EXPECTED.files))
@pytest.mark.parametrize("dtype", ["complex128", "complex64"])
def test_sorted_contraction_indices_on_the_hip_path(name, dtype):
    """The value does not depend on the index order; the plan compiler chooses its own
    layouts either way.  Unsliced cases go through ``array_contract_expression(...,
    sort_contraction_indices=True)`` (reference interface.py:455-456), sliced ones through
    ``tree.sort_contraction_indices(); tree.contract``; and the per-op plug-in with this
    package's own ``einsum`` / ``tensordot`` walks the sorted axes / perms."""
    case = next(c for c in SORTED if c["name"] == name)
    tree = build(case)
    want = EXPECTED[name]
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=case["seed"], dtype="complex128")
    arrays = [a.astype(dtype) for a in arrays]
    tol = 1e-10 if dtype == "complex128" else G.single_gate(want, orc.contract(tree, arrays))
    if not tree.sliced_inds:
        expr = ca.array_contract_expression(
            [tuple(t) for t in case["inputs"]], tuple(case["output"]), case["size_dict"],
            optimize=tree.get_path(), sort_contraction_indices=True,
        )
        assert expr.tree.get_inds_tuple(next(iter(expr.tree.children))) is not None
        plain = build(case)
        plain.sort_contraction_indices()
        assert [expr.tree.get_inds_tuple(p) for p, _, _ in expr.tree.traverse()] == \
               [plain.get_inds_tuple(p) for p, _, _ in plain.traverse()]
        got = np.asarray(expr(*arrays))
        expr.close()
        assert got.shape == want.shape and G.relerr(got, want) <= tol
    tree.sort_contraction_indices()
    got = np.asarray(tree.contract(arrays))
    assert got.shape == want.shape and G.relerr(got, want) <= tol
    got = np.asarray(tree.contract(arrays, implementation=(ca.einsum, ca.tensordot)))
    assert got.shape == want.shape and G.relerr(got, want) <= tol


# ---------------------------------------------------------------------- #
# the fused stem kernel: row-interleaved step 2, intermediates, bf16 x 3 adversarial
# ---------------------------------------------------------------------- #


@pytest.fixture
def fuse_whatever_fits(monkeypatch):
    from cotengra_amd import stem
    monkeypatch.setattr(stem, "gather_rate", lambda run_bytes: 5.4e12)


def stem_names(fn, arrays):
    return [n for n in fn.setup(*arrays)["exec"].step_kernels() if n.startswith("stem2_kernel")]


def pair_with_visible_intermediate(k1, k2, seed):
    """One fused pair  C1 = A . B1  (k1 contracted, k1 new binary indices),  C2 = C1 . B2  with
    B2 = the identity on k2 of C1's indices: C2 IS the intermediate (up to a renaming of
    indices), so the tensor that only ever lives in LDS can be compared element by element."""
    nq = 16 if max(k1, k2) <= 5 else 17
    tree = G.stem_network(nq, [(3, 3), (k1, k1), (k2, k2)], seed)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=seed, dtype="complex64")
    arrays[3] = np.ascontiguousarray(np.eye(2**k2, dtype="complex64").reshape((2,) * (2 * k2)))
    return tree, arrays


def class_errors(got, ref):
    """max and RMS error per magnitude class of the reference (8 binades of max|ref| down):
    a tensor with a wide dynamic range is not judged by its largest elements alone."""
    got, ref = np.asarray(got).reshape(-1), np.asarray(ref).reshape(-1)
    top = np.abs(ref).max()
    out = []
    lo = top
    for _ in range(6):
        hi, lo = lo, lo / 2.0**10
        sel = (np.abs(ref) <= hi) & (np.abs(ref) > lo)
        if sel.sum() < 16:
            continue
        e = np.abs(got[sel] - ref[sel])
        out.append((hi, float(e.max()), float(np.sqrt((e**2).mean()))))
    return out


@pytest.mark.parametrize("k", [(5, 5), (5, 6)])   # k32 n32 | k32 n32, k32 n32 | k64 n64
def test_intermediate_of_a_pair_element_by_element(k, fuse_whatever_fits, monkeypatch):
    """fp32 X / Y form, fp32 row-interleaved form and bf16 x 3 on the SAME pair whose second
    operand is the identity: the intermediate, element by element, against numpy complex128 --
    max and RMS error of every mode within 3 x (max) / 2 x (RMS) of the fp32 X / Y kernel's."""
    # (the random index choice of the network decides between 8- and 16-byte gathers; take the first
    # seed whose pair has a static, row-interleaved instantiation)
    monkeypatch.setenv("CTG_STEM_BF16X3", "0")
    for seed in range(12):
        tree, arrays = pair_with_visible_intermediate(*k, seed=seed)
        fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
        assert sum(s_.kind == KIND_STEM2 for s_ in fn.get_plan("complex64")[0].steps) == 1
        res = {"ri2": np.asarray(fn(*arrays))}
        names = stem_names(fn, arrays)
        if names and G.stem_flags(names[0])["ri2"]:
            break
        fn.close()
    assert names and G.stem_flags(names[0])["ri2"] and not G.stem_flags(names[0])["bf3"], names       # fp32, row-interleaved
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    monkeypatch.setenv("CTG_STEM_NO_RI2", "1")
    res["xy"] = np.asarray(fn(*arrays))
    f_ = G.stem_flags(stem_names(fn, arrays)[0])
    assert not (f_["bf3"] or f_["ri2"] or f_["one"])
    monkeypatch.delenv("CTG_STEM_NO_RI2")
    monkeypatch.delenv("CTG_STEM_BF16X3")          # the default arithmetic: bf16 x 3
    res["bf3"] = np.asarray(fn(*arrays))
    assert G.stem_flags(stem_names(fn, arrays)[0])["bf3"]
    fn.close()
    scale = np.abs(ref).max()
    err = {m: (np.abs(v - ref).max() / scale, np.sqrt((np.abs(v - ref) ** 2).mean()) / scale) for m, v in res.items()}
    print("intermediate k =", k, {m: (f"{a:.2e}", f"{b:.2e}") for m, (a, b) in err.items()})
    assert err["xy"][0] <= 1e-5
    for m in ("ri2", "bf3"):
        assert err[m][0] <= max(3.0 * err["xy"][0], 3e-7), (m, err)
        assert err[m][1] <= max(2.0 * err["xy"][1], 1e-7), (m, err)
    assert not np.array_equal(res["bf3"], res["xy"])


def test_bf16x3_wide_dynamic_range_inside_one_operand(fuse_whatever_fits, monkeypatch):
    """The big operand carries 2^+-60 of dynamic range (every element scaled by its own power of
    two): each value splits into its own three limbs, so nothing is lost relative to ITS
    magnitude -- per magnitude class of the result, the bf16 x 3 error stays within 2 x the fp32
    kernel's own (max within 3 x, RMS within 2 x, or 3e-7 / 1e-7 of the class's top)."""
    nq, gates = G.STEM_CASES[4]           # k32 n32 | k64 n64
    tree = G.stem_network(nq, gates, 404)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=4, dtype="complex64")
    rng = np.random.default_rng(7)
    # the first gate is applied unfused: scale ITS output range through the state tensor, which
    # the first (unfused) step only permutes / mixes over 8 elements
    big = arrays[0]
    arrays[0] = (big * np.exp2(rng.integers(-60, 61, size=big.shape)).astype("float32")).astype("complex64")
    a128 = [a.astype("complex128") for a in arrays]
    ref = np.asarray(orc.contract(tree, a128))
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    monkeypatch.setenv("CTG_STEM_BF16X3", "0")
    fp32 = np.asarray(fn(*arrays))
    monkeypatch.setenv("CTG_STEM_BF16X3", "1")
    bf3 = np.asarray(fn(*arrays))
    assert any(G.stem_flags(n)["bf3"] for n in stem_names(fn, arrays))
    fn.close()
    assert np.isfinite(fp32).all() and np.isfinite(bf3).all()
    c32, c3 = class_errors(fp32, ref), class_errors(bf3, ref)
    assert len(c32) >= 2   # (the contraction mixes all scales: the result spans fewer binades than the operand)
    for (hi, m32, r32), (_, m3, r3) in zip(c32, c3):
        print(f"class <= {hi:.2e}: fp32 max {m32 / hi:.2e} rms {r32 / hi:.2e} | bf16x3 max {m3 / hi:.2e} rms {r3 / hi:.2e}")
        assert m3 <= max(3.0 * m32, 3e-7 * hi), (hi, m3, m32)
        assert r3 <= max(2.0 * r32, 1e-7 * hi), (hi, r3, r32)


def test_bf16x3_hard_cancelling_pair(fuse_whatever_fits, monkeypatch):
    """A big operand that is one constant times (1 + 2^-12 noise) and a small operand whose columns
    sum to zero over the contracted indices: the intermediate (visible through an identity second
    operand) cancels by about 2^-12.  Products are exact in both modes and the accumulation is fp32
    in both: the error RELATIVE TO THE TERMS being summed is the same (2 x, or 3e-7)."""
    k = 5
    tree, arrays = pair_with_visible_intermediate(k, k, seed=11)
    rng = np.random.default_rng(3)
    arrays[1] = np.ascontiguousarray(np.eye(8, dtype="complex64").reshape((2,) * 6))   # first (unfused) gate: identity
    shape = arrays[0].shape
    c = np.complex64(0.7 - 0.4j)
    arrays[0] = (c * (1.0 + 2.0**-12 * rng.normal(size=shape))).astype("complex64")
    # B1 = s(k0) v(k1.., n): the sum over the contracted indices vanishes because sum s = 0
    b1 = arrays[2]
    v = np.repeat(np.take(b1, [0], axis=0), 2, axis=0)
    sign = np.ones((2,) + (1,) * (b1.ndim - 1), dtype="float32")
    sign[1] = -1.0
    arrays[2] = (v * sign).astype("complex64")
    a128 = [a.astype("complex128") for a in arrays]
    ref = np.asarray(orc.contract(tree, a128))
    terms = float(abs(c)) * float(np.abs(arrays[2]).max()) * 2**k      # size of what is being summed
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    monkeypatch.setenv("CTG_STEM_BF16X3", "0")
    fp32 = np.asarray(fn(*arrays))
    monkeypatch.setenv("CTG_STEM_BF16X3", "1")
    bf3 = np.asarray(fn(*arrays))
    assert any(G.stem_flags(n)["bf3"] for n in stem_names(fn, arrays))
    fn.close()
    cancel = np.abs(ref).max() / terms
    e32, e3 = np.abs(fp32 - ref).max() / terms, np.abs(bf3 - ref).max() / terms
    print(f"cancellation {cancel:.2e}; error relative to the terms: fp32 {e32:.2e}, bf16x3 {e3:.2e}")
    assert cancel < 2e-2
    assert e32 <= 2e-6 and e3 <= max(2.0 * e32, 3e-7)


def test_bf16x3_operands_at_the_bottom_of_the_fp32_range(fuse_whatever_fits, monkeypatch):
    """The three-way split is exact while the third limb (2^-16 of the value) is a bf16 number:
    |x| >= 2^-110.  A SMALL operand whose every element is ~2^-112 would lose it (measured before
    the fix: products off by 2e-5) -- the kernel takes a power of two out of a small operand
    whose largest element lies outside [2^-64, 2^64) before splitting it and puts it back in the
    factor its stores apply: the fp32 kernel's accuracy.  The BIG operand cannot be rescaled
    that way (its maximum is not known without a pass over it): a big operand that is tiny as a
    whole -- DOCUMENTED DOMAIN, DESIGN section 4b -- loses its third limb, the products carry a
    relative error of at most 2^-14; ``strip_exponent`` (every stored intermediate renormalised)
    is the answer there, as it is for the fp32 range itself."""
    k = 5
    tree, arrays = pair_with_visible_intermediate(k, k, seed=5)
    tiny = np.float32(2.0**-107)   # (elements of the 1024-element operand are ~2^-5: ~2^-112 after scaling)
    arrays_t = list(arrays)
    arrays_t[2] = (arrays[2] * tiny).astype("complex64")
    # ... and the big operand tiny as a whole: the 2^16-element state (elements ~2^-8) times 2^-104
    # (the first, unfused gate has 8 elements per contraction: the pair's big operand is ~2^-113)
    tiny_a = np.float32(2.0**-104)
    arrays_a = list(arrays)
    arrays_a[0] = (arrays[0] * tiny_a).astype("complex64")
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    monkeypatch.setenv("CTG_STEM_BF16X3", "0")
    fp32_t = np.asarray(fn(*arrays_t)).astype("complex128") / float(tiny)
    fp32_a = np.asarray(fn(*arrays_a)).astype("complex128") / float(tiny_a)
    monkeypatch.setenv("CTG_STEM_BF16X3", "1")
    bf3 = np.asarray(fn(*arrays))
    bf3_t = np.asarray(fn(*arrays_t)).astype("complex128") / float(tiny)
    bf3_a = np.asarray(fn(*arrays_a)).astype("complex128") / float(tiny_a)
    fn.close()
    scale = np.abs(ref).max()
    e = {"fp32, tiny small operand": np.abs(fp32_t - ref).max() / scale, "bf16x3": np.abs(bf3 - ref).max() / scale,
         "bf16x3, tiny small operand": np.abs(bf3_t - ref).max() / scale,
         "fp32, tiny big operand": np.abs(fp32_a - ref).max() / scale,
         "bf16x3, tiny big operand": np.abs(bf3_a - ref).max() / scale}
    print({m: f"{v:.2e}" for m, v in e.items()})
    assert e["fp32, tiny small operand"] <= 1e-5 and e["bf16x3"] <= 1e-5
    assert e["bf16x3, tiny small operand"] <= max(2.0 * e["fp32, tiny small operand"], 1e-6)
    # (round 5: an input that small loses its power of two at upload -- prescale_inputs_kernel -- and the split
    # sees an O(1) operand: the fp32 kernel's own error, no longer the 2^-14 that round 4 documented)
    assert e["bf16x3, tiny big operand"] <= max(2.0 * e["fp32, tiny big operand"], 1e-6)


@pytest.mark.parametrize("case,log2_scale,underflows", [(10, -16, True), (0, -12, False), (4, -12, False)])
def test_bf16x3_unrescaled_inputs_under_strip_exponent(case, log2_scale, underflows, fuse_whatever_fits, monkeypatch):
    """Frobenius-normalised (un-rescaled) inputs, scaled down further by 2^-12 ... 2^-16 each,
    contracted with ``strip_exponent``: every stored intermediate is renormalised by its
    producer's max|.|, so the operands of a pair are O(1) x (raw input) and mantissa x
    10^exponent matches the oracle in both arithmetics -- on the long stem (7 gates) although
    the value itself is far below the fp32 range."""
    nq, gates = G.STEM_CASES[case]
    tree = G.stem_network(nq, gates, 100 * case)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=case, dtype="complex64")
    arrays = [(a * np.float32(2.0**log2_scale)).astype("complex64") for a in arrays]
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    assert (0 < np.abs(ref).max() < 1e-38) == underflows     # (the plain fp32 result would underflow)
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    out = {}
    monkeypatch.setenv("CTG_STEM_BF16X3", "0")
    m, e = fn(*arrays, strip_exponent=True)
    out["fp32"] = np.asarray(m).astype("complex128") * 10.0**e
    monkeypatch.setenv("CTG_STEM_BF16X3", "1")
    m, e = fn(*arrays, strip_exponent=True)
    out["bf16x3"] = np.asarray(m).astype("complex128") * 10.0**e
    assert any(G.stem_flags(n)["bf3"] for n in stem_names(fn, arrays))
    fn.close()
    err = {k_: G.relerr(v, ref) for k_, v in out.items()}
    print(err)
    assert err["fp32"] <= 1e-5 and err["bf16x3"] <= max(1e-5, 2.0 * err["fp32"])


# ---------------------------------------------------------------------- #
# single stem steps: the stem kernel's first half alone (csrc/ctg_stem.hip: ONE)
# ---------------------------------------------------------------------- #

from test_host_round4 import ONE_CASES  # noqa: E402


@pytest.fixture
def take_every_single(monkeypatch):
    from cotengra_amd import stem
    monkeypatch.setattr(stem, "gather_rate", lambda run_bytes: 5.4e12)
    monkeypatch.setattr(stem, "MIN_GAIN", -1e9)


@pytest.mark.parametrize("case", range(len(ONE_CASES)))
@pytest.mark.parametrize("sliced", [0, 2])
def test_single_stem_steps_on_the_device(case, sliced, take_every_single, monkeypatch):
    """A large step no pair took runs on the stem kernel's first half: fp32 products (static
    instantiation and the run-time-count variant: the same bits), bf16 x 3 (the default), with and
    without ``strip_exponent``, slice by slice -- against the numpy complex128 oracle."""
    nq, gates = ONE_CASES[case]
    tree = G.stem_network(nq, gates, 300 + case, sliced=sliced)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=case, dtype="complex64")
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    gate = G.single_gate(ref, orc.contract(tree, arrays))
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    ones = [s_ for s_ in fn.get_plan("complex64")[0].steps if s_.kind == KIND_STEM2 and s_.stem.get("one")]
    if sliced and not ones:
        pytest.skip("the sliced indices are contracted ones of the step")
    assert len(ones) == 1
    got = {}
    got["default"] = np.asarray(fn(*arrays))
    # (a single step's kernel: the last three template arguments are BF3, RI2 = false, ONE = true)
    one_names = [n for n in stem_names(fn, arrays) if G.stem_flags(n)["one"]]
    assert len(one_names) == 1, stem_names(fn, arrays)
    one_name = one_names[0]
    monkeypatch.setenv("CTG_STEM_BF16X3", "0")
    got["fp32"] = np.asarray(fn(*arrays))
    m, e = fn(*arrays, strip_exponent=True)
    got["fp32 strip"] = np.asarray(m).astype("complex128") * 10.0**e
    monkeypatch.setenv("CTG_STEM_GENERIC", "1")
    got["fp32 generic"] = np.asarray(fn(*arrays))
    gnames = stem_names(fn, arrays)
    monkeypatch.delenv("CTG_STEM_GENERIC")
    monkeypatch.setenv("CTG_STEM_BF16X3", "1")
    got["bf16x3"] = np.asarray(fn(*arrays))
    bnames = stem_names(fn, arrays)
    m, e = fn(*arrays, strip_exponent=True)
    got["bf16x3 strip"] = np.asarray(m).astype("complex128") * 10.0**e
    if tree.nslices > 1:
        a128 = [a.astype("complex128") for a in arrays]
        for i in range(tree.nslices):
            ri = np.asarray(orc.contract_slice(tree, a128, i))
            gi = max(gate, G.single_gate(ri, orc.contract_slice(tree, arrays, i)))
            assert G.relerr(np.asarray(fn.contract_slice(arrays, i)), ri) <= gi
    fn.close()
    print(one_name, {k_: f"{G.relerr(v, ref):.2e}" for k_, v in got.items()})
    for k_, v in got.items():
        assert G.relerr(v, ref) <= gate, (k_, G.relerr(v, ref), gate)
    assert np.array_equal(got["fp32"], got["fp32 generic"])
    assert any(f["one"] and f["nch"] == 0 and not f["bf3"] for f in map(G.stem_flags, gnames)), gnames
    assert np.array_equal(got["default"], got["bf16x3"])
    if any(f["one"] and f["bf3"] for f in map(G.stem_flags, bnames)):   # (static: it ran on the bf16 pipe)
        assert not np.array_equal(got["bf16x3"], got["fp32"])


# ---------------------------------------------------------------------- #
# two GPUs, when the box has them (the driver's 8-GPU node; skipped on the one-GPU lease)
# ---------------------------------------------------------------------- #


def _n_gpus():
    import torch

    return torch.cuda.device_count()


def test_two_ranks_plain_c_driver_and_bench_line(tmp_path):
    """``tests/cabi_reduce`` with world = 2 (no Python in the ranks: plan file -> executors ->
    RCCL unique id through a file -> ``ctg_exec_run_share(rank, 2)`` ->
    ``ctg_exec_reduce``) and ``bench.py --gpus 2 --headline-only``: two distinct devices, the
    reduce behind the C ABI, the 2-rank amplitude equal to the 1-rank one within the
    single-precision gate, and the per-rank slice times (load balance) printed."""
    if _n_gpus() < 2:
        pytest.skip("one GPU visible: the multi-rank RCCL path needs two (covered by gloo world-2 CPU tests "
                    "and the one-rank RCCL tests)")
    binary = os.path.join(ROOT, "tests", "cabi_reduce")
    plan = os.path.join(ROOT, "tests", "golden", "cabi_plan.bin")
    idf = str(tmp_path / "id")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([binary, plan, str(r), "2", idf], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True, env=env) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK" in outs[0], outs

    def bench(n):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
               "--headline-only", "--no-cpu-baseline"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        line = r.stdout.strip().splitlines()[-1]
        assert len(line) < 4096                       # (the compact record; the full one is a file)
        return json.load(open(os.path.join(ROOT, json.loads(line)["full_record"])))

    two, one = bench(2), bench(1)
    assert two["n_gpus"] == 2 and two["distinct_gpus"] == 2
    assert two["config"]["reduce_via"] == "ctg_exec_reduce (RCCL, C ABI)"
    spread = [r["slices_ms"] for r in two["per_rank"]]
    print("per-rank slices_ms:", spread, "reduce wait:", [r["reduce_wait_ms"] for r in two["per_rank"]])
    assert max(spread) <= 1.25 * min(spread)
    # weak scaling: twice the slices in (about) the same time
    assert two["value"] >= 1.6 * one["value"]
    # the amplitudes: rank 0 of the 2-rank run holds the sum of slices {2, 3} + {4, 5}... of its own
    # schedule; compare like with like -- 4 slices on one rank
    cmd1 = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
            "--headline-only", "--no-cpu-baseline"]
    r1 = subprocess.run(cmd1, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r1.returncode == 0, r1.stderr[-2000:]
    full1 = json.load(open(os.path.join(ROOT, json.loads(r1.stdout.strip().splitlines()[-1])["full_record"])))
    a1 = complex(*full1["config"]["partial_amplitude"])
    a2 = complex(*two["config"]["partial_amplitude"])
    assert abs(a2 - a1) <= 1e-4 * abs(a1), (a1, a2)


# ---- three-step tiles (opt-in: CTG_STEM_TRIPLES) -----------------------------------------------------------

@pytest.mark.parametrize("bf16x3", [True, False])
@pytest.mark.parametrize("seed", (0, 27, 34, 41, 45))
def test_three_step_tiles_on_device(seed, bf16x3, monkeypatch):
    """Random stems planned with a three-step tile (the shapes csrc/ctg_stem.hip: CTG_STEM_TRI holds for
    them), run in both arithmetics -- the plan is made for bf16 x 3, the fp32 kernels of the same shapes
    serve an executor whose arithmetic is switched afterwards -- against the complex128 oracle within the
    gate of the unfused HIP path; the kernel that ran carries a middle stage (its last two template
    arguments)."""
    from cotengra_amd import runtime, stem

    if runtime.load().ctg_stem_triple_instantiated(1, 0, 1, 2, 1, 1, 1, 1, 0) != 1:
        pytest.skip("library built without three-step tiles (round 5: slower than pairs on every tree; "
                    "tools/build_variants.py triples=-DCTG_STEM_TRIPLES_BUILD)")
    monkeypatch.setenv("CTG_STEM_TRIPLES", "1")
    monkeypatch.delenv("CTG_STEM_BF16X3", raising=False)
    monkeypatch.setattr(stem, "gather_rate", lambda run_bytes: 5.4e12)
    monkeypatch.setattr(stem, "TRIPLE_STAGE_RATE", {n: 1e15 for n in stem.TRIPLE_STAGE_RATE})
    tree = G.random_stem(seed)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=seed, dtype="complex64")
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    plain = HipContractor(tree, fuse=False)
    base = np.asarray(plain(*arrays))
    plain.close()
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 9, stem_bf16x3=True)
    plan = fn.get_plan("complex64")[0]
    assert [s for s in plan.steps if s.kind == KIND_STEM2 and s.stem.get("KM")]
    fn.stem_bf16x3 = bf16x3          # (the executor's arithmetic; the plan stays)
    got = np.asarray(fn(*arrays))
    names = [n for n in fn.setup(*arrays)["exec"].step_kernels() if n.startswith("stem2_kernel<") and n.count(",") == 13]
    fn.close()
    assert names and all(n.split(",")[9] == ("true" if bf16x3 else "false") and int(n.split(",")[12]) >= 1 for n in names), names
    scale = np.abs(ref).max()
    gate = max(1e-5, 8 * np.abs(base - ref).max() / scale)
    assert np.abs(got - ref).max() / scale <= gate


# ---- slice groups ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("fixture", ["sycamore_m20_w32_r4.json", "sycamore_m20_native.json"])
def test_slice_groups_narrowed_against_oracle(fixture, monkeypatch):
    """m20 trees narrowed to width 2^20 with slice groups in the plan: whole groups, a partial group and
    lone slices in one ``run_slice_list`` call (in descending order: the executor sorts them group by
    group) against the complex128 oracle, within the gate numpy's own single precision sets."""
    from cotengra_amd import plan as P
    from oracle.plan_interp import group_members

    monkeypatch.delenv("CTG_SLICE_GROUPS", raising=False)
    monkeypatch.setattr(P, "GROUP_MIN_WIDTH", 1)
    monkeypatch.setattr(P, "GROUP_MIN_SAVING", 0.0)
    tree = ca.tree_from_record(ca.load_network(os.path.join(os.path.dirname(__file__), "golden", "trees", fixture)))
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=42, dtype="complex64", rescale=True)
    small = tree.slice(target_size=2**20)
    fn = HipContractor(small)
    plan = fn.get_plan("complex64")[0]
    assert plan.group_size >= 2 and any(s.group for s in plan.steps)
    ids = group_members(plan, 3) + group_members(plan, 77777)[:3] + [12345, 5]
    a128 = [a.astype("complex128") for a in arrays]
    ref = sum(complex(orc.contract_slice(small, a128, i)) for i in ids)
    np64 = sum(complex(orc.contract_slice(small, arrays, i)) for i in ids)
    ex = fn.setup(*arrays)["exec"]
    ex.zero_result()
    ex.run_slice_list(ids[::-1])
    got = complex(ex.download_result())
    # ... and the same slices one call each (nothing shared between calls but what the key says is there)
    ex.zero_result()
    for i in sorted(ids, key=lambda i: (plan.group_of(i), i)):
        ex.run_slice_list([i])
    again = complex(ex.download_result())
    fn.close()
    assert abs(got - ref) / abs(ref) <= max(1e-5, 8.0 * abs(np64 - ref) / abs(ref))
    assert again == got


def test_slice_groups_full_width_bit_identical(monkeypatch):
    """One group of the time-to-solution tree at full width: the shared steps computed once == every step
    computed for every slice, bit for bit (same kernels, same order of additions)."""
    tree = ca.tree_from_record(ca.load_network(os.path.join(os.path.dirname(__file__), "golden", "trees",
                                                            "sycamore_m20_w32_r4.json")))
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=42, dtype="complex64", rescale=True)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CTG_SLICE_GROUPS", mode)
        fn = HipContractor(tree)
        plan = fn.get_plan("complex64")[0]
        if mode == "1":
            assert plan.group_size == 4
            members = plan.group_ids(plan.group_of(5))
        else:
            assert plan.group_size == 1
        ex = fn.setup(*arrays)["exec"]
        ex.zero_result()
        ex.run_slice_list(members)
        res[mode] = complex(ex.download_result())
        fn.close()
    assert res["1"] == res["0"] and res["1"] != 0


GROUP_CASES = [c["name"] for c in G.cases("tree")
               if c["name"] in ("lattice4x4_sliced", "lattice8x8_sliced", "preproc_s0_ac", "preproc_s1_ac")
               or (c["name"].startswith("rand_") and c["name"].endswith("sliced"))]


@pytest.fixture
def groups_everywhere(monkeypatch):
    from cotengra_amd import plan as P

    monkeypatch.delenv("CTG_SLICE_GROUPS", raising=False)
    monkeypatch.setattr(P, "GROUP_MIN_WIDTH", 1)
    monkeypatch.setattr(P, "GROUP_MIN_SAVING", 0.0)


@pytest.mark.parametrize("name", GROUP_CASES)
def test_slice_groups_on_golden_trees(name, groups_everywhere):
    """Every sliced golden tree (hyper and output indices sliced, extents 2 and 3, single-term
    preprocessing) contracted with slice groups forced on: the whole contraction against the oracle."""
    case = next(c for c in G.cases("tree") if c["name"] == name)
    tree = G.tree_of(case)
    if tree.multiplicity < 4:
        pytest.skip("fewer than four slices")
    arrays = [a.astype("complex64") for a in G.arrays_of(case, "complex128", tree)]
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    fn = HipContractor(tree)
    if fn.get_plan("complex64")[0].group_size < 2:
        fn.close()
        pytest.skip("no step is independent of a sliced index")
    got = np.asarray(fn(*arrays))
    fn.close()
    tol = G.single_gate(ref, orc.contract(tree, arrays)) * np.abs(ref).max()
    assert np.abs(got - ref).max() <= tol


def test_slice_groups_resumable_and_list_api(tmp_path, groups_everywhere):
    """With slice groups in the plan the resumable contraction sums this rank's slices group by group, in
    chunks of whole groups, and a run killed in between resumes to the same bits; run_slice_list takes
    repetitions (a slice given twice is added twice) and refuses ids outside the tree."""
    from cotengra_amd import runtime
    from cotengra_amd.contractor import contract_resumable

    case = next(c for c in G.cases("tree") if c["name"] == "rand_s42_r3_o1_hi0_ho0_outsliced")
    tree = G.tree_of(case)
    arrays = [a.astype("complex64") for a in G.arrays_of(case, "complex128", tree)]
    fn = HipContractor(tree)
    plan = fn.get_plan("complex64")[0]
    assert plan.group_size >= 2 and tree.multiplicity >= 4 * plan.group_size
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    ck = str(tmp_path / "ck.npz")
    assert contract_resumable(tree, arrays, ck, every=3, stop_after=2 * plan.group_size) is None
    assert os.path.exists(ck)
    out = np.asarray(contract_resumable(tree, arrays, ck, every=3))
    whole = np.asarray(contract_resumable(tree, arrays, str(tmp_path / "ck2.npz"), every=3))
    assert np.array_equal(out, whole)
    scale = np.abs(ref).max()
    assert np.abs(out - ref).max() <= G.single_gate(ref, orc.contract(tree, arrays)) * scale
    ex = fn.setup(*arrays)["exec"]
    ex.zero_result()
    ex.run_slice_list([1, 1, 0])
    twice = np.asarray(ex.download_result()).copy()
    ex.zero_result()
    ex.run_slice_list([0])
    ex.run_slice_list([1])
    ex.run_slice_list([1])
    assert np.allclose(twice, np.asarray(ex.download_result()), rtol=1e-5, atol=1e-6 * scale)
    ex.run_slice_list([])
    with pytest.raises((runtime.CtgError, ValueError)):
        ex.run_slice_list([tree.multiplicity])
    fn.close()
