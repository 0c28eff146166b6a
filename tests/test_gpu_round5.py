"""GPU suite, round 5.

* a SHARED leaf-preprocessing step feeding a per-slice step in batched slice-group launches (the
  advisor's round-4 finding: the consumer named no producer and read an arena replica nobody wrote);
* a rank's share of the slices through ``ctg_exec_run_share`` (whole slice groups rank, rank + world,
  ...): the shares of all ranks add up to the contraction, for trees with and without groups;
* two GPUs, when the box has them: ``tree.contract_mpi`` over RCCL.
"""
import os
import socket
import sys

import numpy as np
import pytest

import cotengra_amd as ca
from cotengra_amd.contractor import HipContractor
from oracle import contract_ref as orc

import golden_util as G
from test_host_round5 import shared_single_tree

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dtype", ["complex128", "complex64"])
@pytest.mark.parametrize("extra", [(), ("e",), ("q",)])
def test_shared_single_step_feeds_a_per_slice_step(dtype, extra, monkeypatch):
    """Default planner thresholds: group index g, launches carry whole groups (z = group * d + member),
    the preprocessed leaf (a, b, z) -> (a, b) is shared by the two slices of a group and read by a step
    that is not.  Against the oracle, and bit-identical to the run that launches slice by slice."""
    monkeypatch.delenv("CTG_SLICE_GROUPS", raising=False)
    tree = shared_single_tree(extra)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=11, dtype="complex128")
    ref = complex(orc.contract(tree, arrays))
    arrays = [a.astype(dtype) for a in arrays]
    tol = 1e-10 if dtype == "complex128" else G.single_gate(ref, orc.contract(tree, arrays))
    fn = HipContractor(tree)
    plan = fn.get_plan(dtype)[0]
    assert plan.group_size >= 2 and plan.steps[0].kind == 0
    if not extra:     # (the advisor's tree itself: the preprocessing step is shared, its consumer is not)
        assert plan.steps[0].group and plan.group_inds == ("g",)
    ex = fn.setup(*arrays)["exec"]
    assert ex.batch >= plan.group_size          # batched launches of whole groups
    got = complex(np.asarray(fn(*arrays)))
    fn.close()
    assert abs(got - ref) <= tol * abs(ref), (got, ref)
    monkeypatch.setenv("CTG_NO_BATCHED_GROUPS", "1")
    fn1 = HipContractor(tree)
    one = complex(np.asarray(fn1(*arrays)))
    fn1.close()
    assert abs(one - ref) <= tol * abs(ref)
    # every slice on its own: the shared step is computed for each
    fn2 = HipContractor(tree)
    ex2 = fn2.setup(*arrays)["exec"]
    ex2.zero_result()
    for i in range(tree.nslices):
        ex2.run_slice_list([i])
    lone = complex(np.asarray(ex2.download_result()))
    fn2.close()
    assert abs(lone - ref) <= tol * abs(ref)


SHARE_CASES = ["lattice8x8_sliced", "rand_s42_r3_o1_hi0_ho0_outsliced", "rand_s42_r2_o2_hi0_ho2_outsliced",
               "C5_hyper200", "preproc_s1_ac"]


@pytest.mark.parametrize("name", SHARE_CASES)
@pytest.mark.parametrize("groups", [True, False])
def test_shares_of_all_ranks_add_up(name, groups, monkeypatch):
    """``ctg_exec_run_share(rank, world)`` for every rank of a world of 1, 2, 3 and 5 into one result
    tensor = the contraction (each slice exactly once); in units (a checkpointing caller's view) too."""
    from cotengra_amd import plan as P

    if groups:
        monkeypatch.delenv("CTG_SLICE_GROUPS", raising=False)
        monkeypatch.setattr(P, "GROUP_MIN_WIDTH", 1)
        monkeypatch.setattr(P, "GROUP_MIN_SAVING", 0.0)
        monkeypatch.setattr(P, "GROUP_MIN_SAVING_SMALL", 0.0)
    else:
        monkeypatch.setenv("CTG_SLICE_GROUPS", "0")
    case = next(c for c in G.cases("tree") if c["name"] == name)
    tree = G.tree_of(case)
    arrays = G.arrays_of(case, "complex128", tree)
    ref = np.asarray(orc.contract(tree, arrays)) if tree.nslices <= 4096 else np.ones(1)
    fn = HipContractor(tree)
    st = fn.setup(*arrays)
    ex, plan = st["exec"], st["plan"]
    if groups and plan.group_size < 2:
        fn.close()
        pytest.skip("no step is independent of a sliced index")
    scale = np.abs(ref).max()
    if tree.nslices > 4096:
        # (C5: 3.9e9 slices -- windows of a few units of every rank's share against the oracle's slices)
        from cotengra_amd.distributed import scatter_slices

        a128 = [np.asarray(a).astype("complex128") for a in arrays]
        for world in (3, 8):
            ex.zero_result()
            want = 0.0
            for rank in range(world):
                units, gs = plan.share_units(rank, world)
                u0 = units - 2 if rank % 2 else 1
                ex.run_share(rank, world, u0, 2)
                ids = plan.rank_slice_ids(rank, world, u0, 2)
                assert len(ids) == 2 * gs and len({int(plan.group_of(int(i))) for i in ids}) == 2
                # (sliced OUTPUT indices: every slice lands in its own chunk of the result)
                want = want + scatter_slices(tree, [int(i) for i in ids],
                                             [np.asarray(orc.contract_slice(tree, a128, int(i))) for i in ids])
            got = np.asarray(ex.download_result())
            assert got.shape == want.shape and np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
        fn.close()
        return
    for world in (1, 2, 3, 5):
        ex.zero_result()
        for rank in range(world):
            units, gs = plan.share_units(rank, world)
            if rank % 2 or units > 128:
                ex.run_share(rank, world)
            else:               # unit by unit, from the back
                for u in reversed(range(units)):
                    ex.run_share(rank, world, u, 1)
        got = np.asarray(ex.download_result())
        assert np.abs(got - ref).max() <= 1e-10 * scale, (world, np.abs(got - ref).max() / scale)
    # beyond the share: refused
    units, _ = plan.share_units(0, 2)
    with pytest.raises(ValueError):
        ex.run_share(0, 2, units, 1)
    with pytest.raises(ValueError):
        ex.run_share(2, 2)
    fn.close()


def test_whole_tree_call_uses_the_share_path(monkeypatch):
    """``fn(*arrays)`` = the share of rank 0 of 1, also under a progress bar (chunks of whole units)."""
    case = next(c for c in G.cases("tree") if c["name"] == "rand_s42_r3_o1_hi0_ho0_outsliced")
    tree = G.tree_of(case)
    arrays = G.arrays_of(case, "complex128", tree)
    ref = np.asarray(orc.contract(tree, arrays))
    fn = HipContractor(tree)
    a = np.asarray(fn(*arrays))
    b = np.asarray(fn(*arrays, progbar=True))
    fn.close()
    assert np.abs(a - ref).max() <= 1e-10 * np.abs(ref).max()
    assert np.abs(b - ref).max() <= 1e-10 * np.abs(ref).max()


# ---------------------------------------------------------------------- #
# two GPUs: tree.contract_mpi over RCCL (skipped on the one-GPU lease)
# ---------------------------------------------------------------------- #


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _mpi_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from cotengra_amd import plan as P
        from cotengra_amd.distributed import close_comms

        P.GROUP_MIN_WIDTH, P.GROUP_MIN_SAVING, P.GROUP_MIN_SAVING_SMALL = 1, 0.0, 0.0
        ok = True
        for name in ("rand_s42_r3_o1_hi0_ho0_outsliced", "rand_s42_r2_o2_hi0_ho2_outsliced", "lattice8x8_sliced"):
            case = next(c for c in G.cases("tree") if c["name"] == name)
            tree = G.tree_of(case)
            arrays = G.arrays_of(case, "complex128", tree)
            ref = np.asarray(orc.contract(tree, arrays))
            if tree.sliced_inds and not any(ix in tree.output for ix in tree.sliced_inds):
                out = tree.contract_mpi(arrays)                      # all-reduce: every rank holds the result
                ok = ok and np.abs(np.asarray(out) - ref).max() <= 1e-10 * np.abs(ref).max()
                out = tree.contract_mpi(arrays, root=1)
                ok = ok and ((out is None) if rank != 1 else np.abs(np.asarray(out) - ref).max() <= 1e-10 * np.abs(ref).max())
            out = tree.contract_distributed(arrays)                  # (takes sliced output indices too)
            ok = ok and np.abs(np.asarray(out) - ref).max() <= 1e-10 * np.abs(ref).max()
        close_comms()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_contract_mpi_over_rccl_two_gpus():
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: tree.contract_mpi over RCCL needs two (the partition is covered by the "
                    "CPU tests, the reduce with one rank by test_gpu_round3)")
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mpi_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    results = sorted(q.get(timeout=10) for _ in procs)
    assert results == [(0, True), (1, True)]


# ---------------------------------------------------------------------- #
# the slices are summed in double precision (accum_kernel's wide running sum)
# ---------------------------------------------------------------------- #


def test_slices_are_summed_in_double_precision(monkeypatch):
    """1024 slices of an m20 tree narrowed to CPU size.  The per-slice values the device adds are fetched
    one by one (``contract_slice``: the same kernels, bit for bit); their exact sum in float64 is what the
    device's total must be after ONE rounding -- the complex64 left fold of the same values (what
    ``gather_slices``, core.py:3842-3844, and this executor until round 4 did) is measurably further off.
    The state of an interrupted run is the double-precision sum: get / set round-trip to the same bits."""
    from test_tree_fixtures import narrowed

    monkeypatch.setenv("CTG_SLICE_GROUPS", "0")     # (one slice at a time must be the same arithmetic as all at once)
    tree = ca.tree_from_record(ca.load_network(os.path.join(ROOT, "tests", "golden", "trees", "sycamore_m20_w32_r4.json")))
    small = narrowed(tree, 10)
    arrays = ca.make_arrays_from_inputs(small.inputs, small.size_dict, seed=42, dtype="complex64", rescale=True)
    n = 1024
    fn = HipContractor(small, handle_slicing=True)
    st = fn.setup(*arrays)
    ex = st["exec"]
    assert ex.state_dtype() == np.dtype("complex128")
    vals = np.empty(n, dtype=np.complex64)
    for i in range(n):
        ex.zero_result()
        ex.run_slices(i, 1, 1)
        vals[i] = ex.download_result()
    exact = vals.astype(np.complex128).sum()
    fold32 = np.complex64(0)
    for v in vals:
        fold32 = np.complex64(fold32 + v)
    ex.zero_result()
    ex.run_slices(0, n, 1)
    dev = complex(ex.download_result())
    wide, _, _ = ex.get_state_wide()
    err_dev, err_fold = abs(dev - exact) / abs(exact), abs(complex(fold32) - exact) / abs(exact)
    print(f"sum of {n} slices: device {err_dev:.2e}, complex64 left fold {err_fold:.2e} (relative to the exact sum)")
    assert abs(complex(wide) - exact) <= 1e-12 * abs(exact)
    assert err_dev <= 1.2e-7                       # one rounding to single precision
    assert err_fold >= 4.0 * err_dev               # the fp32 running sum is measurably worse
    # resume: half, state out, state in (a fresh sum), other half -> the same bits as in one go
    ex.zero_result()
    ex.run_slices(0, n // 2, 1)
    mid, e_, z_ = ex.get_state_wide()
    assert mid.dtype == np.complex128
    ex.zero_result()
    ex.set_state(mid, e_, z_)
    ex.run_slices(n // 2, n // 2, 1)
    again, _, _ = ex.get_state_wide()
    assert complex(again) == complex(wide) and complex(ex.download_result()) == dev
    # against the oracle: a 256-slice prefix in complex128
    ex.zero_result()
    ex.run_slices(0, 256, 1)
    got = complex(ex.download_result())
    a128 = [a.astype("complex128") for a in arrays]
    ref = sum(complex(orc.contract_slice(small, a128, i)) for i in range(256))
    assert abs(got - ref) <= 1e-5 * abs(ref)
    fn.close()
    # the switch: CTG_NO_WIDE_SUM=1 is the round-4 arithmetic
    monkeypatch.setenv("CTG_NO_WIDE_SUM", "1")
    fn2 = HipContractor(small, handle_slicing=True)
    ex2 = fn2.setup(*arrays)["exec"]
    assert ex2.state_dtype() == np.dtype("complex64")
    ex2.zero_result()
    ex2.run_slices(0, n, 1)
    old = complex(ex2.download_result())
    fn2.close()
    assert abs(old - exact) / abs(exact) >= err_dev


def test_outer_sliced_and_real_trees_keep_their_results_with_the_wide_sum():
    """float32 and complex64 golden trees whose sliced indices are output indices (every slice lands in its own
    chunk of the result) and inner-sliced ones: against the oracle, and the wide state has the result's shape."""
    for name in ("rand_s42_r3_o1_hi0_ho0_outsliced", "lattice8x8_sliced", "rand_s42_r2_o2_hi0_ho2_outsliced"):
        case = next(c for c in G.cases("tree") if c["name"] == name)
        tree = G.tree_of(case)
        for dtype in (("complex64", "float32") if "ho2" not in name else ("complex64",)):   # (the oracle itself refuses the real parts of the hyper-output tree)
            arrays = G.arrays_of(case, "complex128", tree)
            if dtype == "float32":
                arrays = [np.ascontiguousarray(a.real) for a in arrays]
            ref = np.asarray(orc.contract(tree, [a.astype("float64" if dtype == "float32" else "complex128") for a in arrays]))
            arrays = [a.astype(dtype) for a in arrays]
            fn = HipContractor(tree)
            got = np.asarray(fn(*arrays))
            st = fn.setup(*arrays)
            wide = st["exec"].state_dtype()
            fn.close()
            assert wide == np.dtype("float64" if dtype == "float32" else "complex128")
            tol = G.single_gate(ref, orc.contract(tree, arrays)) * np.abs(ref).max()
            assert got.shape == ref.shape and np.abs(got - ref).max() <= tol, (name, dtype)


# ---------------------------------------------------------------------- #
# input tensors far from 1: an exact power of two comes out at upload (prescale_inputs_kernel)
# ---------------------------------------------------------------------- #


@pytest.fixture
def fuse_whatever_fits(monkeypatch):
    from cotengra_amd import stem
    monkeypatch.setattr(stem, "gather_rate", lambda run_bytes: 5.4e12)


@pytest.mark.parametrize("bf16x3", ["1", "0"])
def test_inputs_at_2_to_the_minus_45_under_strip_exponent(bf16x3, fuse_whatever_fits, monkeypatch):
    """The 7-gate stem with every raw (Frobenius-normalised) input scaled by 2^-45: the value, ~2^-400, exists
    only as mantissa x 10^exponent.  The reference normalises after every step (contract.py:816-829); this
    executor normalises lazily, and two such inputs meeting in one fused pair used to underflow the fp32
    intermediate before any scale was applied (round 4: DOCUMENTED; the test then stopped at 2^-16).  The
    inputs now lose their power of two at upload (exact) and the exponent gets it back."""
    monkeypatch.setenv("CTG_STEM_BF16X3", bf16x3)
    nq, gates = G.STEM_CASES[10]
    tree = G.stem_network(nq, gates, 1000)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=10, dtype="complex64")
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    shift = -45 * len(arrays)
    scaled = [(a * np.float32(2.0**-45)).astype("complex64") for a in arrays]
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    m, e = fn(*scaled, strip_exponent=True)
    names = [n for n in fn.setup(*scaled)["exec"].step_kernels() if n.startswith("stem2_kernel")]
    fn.close()
    assert names
    # mantissa x 10^e = ref x 2^shift: compare in logarithms (the value is far below every float range)
    got_log10 = np.log10(np.abs(np.asarray(m).astype("complex128")).max()) + e
    want_log10 = np.log10(np.abs(ref).max()) + shift * np.log10(2.0)
    assert abs(got_log10 - want_log10) <= 1e-5
    mant = np.asarray(m).astype("complex128") / np.abs(np.asarray(m)).max()
    assert np.abs(mant - ref / np.abs(ref).max()).max() <= 1e-5


@pytest.mark.parametrize("bf16x3", ["1", "0"])
def test_inputs_at_both_ends_of_the_fp32_range(bf16x3, fuse_whatever_fits, monkeypatch):
    """Inputs alternately scaled by 2^+50 and 2^-50 (and the big state by 2^-104: the round-4 'tiny big
    operand', which cost the bf16 x 3 split its third limb -- accepted then up to 2^-14): the value is an
    ordinary number, every un-prescaled fp32 intermediate would overflow or underflow on the way.  Both
    arithmetics at the fp32 kernel's own accuracy; inputs inside [2^-32, 2^32) are not touched (same bits
    as with the pass switched off)."""
    monkeypatch.setenv("CTG_STEM_BF16X3", bf16x3)
    nq, gates = G.STEM_CASES[10]
    tree = G.stem_network(nq, gates, 1000)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=10, dtype="complex64", rescale=True)
    ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
    scale = np.abs(ref).max()
    fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    plain = np.asarray(fn(*arrays))
    e_plain = np.abs(plain - ref).max() / scale
    # (an even number of small inputs: the powers cancel)
    n_small = (len(arrays) - 1) // 2 * 2
    powers = [0] + [50 if i % 2 else -50 for i in range(n_small)] + [0] * (len(arrays) - 1 - n_small)
    ends = [(a * np.float32(2.0**p)).astype("complex64") for a, p in zip(arrays, powers)]
    got = np.asarray(fn(*ends))
    assert np.array_equal(got, plain)               # exact powers of two out and back in: the same bits
    tiny = list(arrays)
    tiny[0] = (arrays[0] * np.float32(2.0**-104)).astype("complex64")
    got_t = np.asarray(fn(*tiny)).astype("complex128") * 2.0**104
    e_tiny = np.abs(got_t - ref).max() / scale
    fn.close()
    print(f"plain {e_plain:.2e}, big operand at 2^-104: {e_tiny:.2e}")
    assert e_plain <= 1e-5 and e_tiny <= max(2.0 * e_plain, 1e-6)
    monkeypatch.setenv("CTG_NO_PRESCALE", "1")
    fn0 = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
    off = np.asarray(fn0(*arrays))
    fn0.close()
    assert np.array_equal(off, plain)


# ---------------------------------------------------------------------- #
# long tiled steps with bf16 x 3 products (csrc/ctg_pair_mfma.hip: pair_mfma_bf3_kernel)
# ---------------------------------------------------------------------- #


def _gemm_tree(R, K, N):
    import cotengra_amd as ca_

    return ca_.ContractionTree.from_path([("a", "b"), ("b", "c")], ("a", "c"), dict(a=R, b=K, c=N), path=[(0, 1)])


@pytest.mark.parametrize("R,K,N", [(8192, 512, 512), (65536, 64, 64), (16384, 256, 128), (1024, 512, 512)])
def test_long_tiled_steps_multiply_on_the_bf16_pipe(R, K, N, monkeypatch):
    """A GEMM-like complex64 step with K >= 64 on full 64-column tiles runs pair_mfma_bf3_kernel (fp32 operands
    split exactly into three ROUNDED bf16 limbs where they are staged into LDS, six products on
    v_mfma_f32_32x32x16_bf16) unless CTG_PAIR_BF16X3 / CTG_STEM_BF16X3 = 0.  Against the complex128 oracle:
    the fp32 kernel's accuracy -- random data, 2^+-30 of dynamic range across rows and columns (judged against
    the size of each element's own terms), and a contraction that cancels by 2^-12.  A step too small to fill the
    chip with 64-column tiles keeps fp32 products whatever the switch says."""
    monkeypatch.delenv("CTG_STEM_BF16X3", raising=False)
    tree = _gemm_tree(R, K, N)
    rng = np.random.default_rng(R + K + N)

    def cplx(*shape):
        return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype("complex64")

    a, b = cplx(R, K), cplx(K, N)
    eligible = (R // 128) * (N // 64) >= 512     # (enough 64-column tiles to fill the chip: MfmaHints::bf3)
    row = np.exp2(rng.integers(-30, 31, size=R)).astype("float32")
    col = np.exp2(rng.integers(-30, 31, size=N)).astype("float32")
    a_wide, b_wide = (a * row[:, None]).astype("complex64"), (b * col[None, :]).astype("complex64")
    a_canc = a.copy()
    a_canc[:, K // 2:] = -a[:, : K // 2]
    b_canc = b.copy()
    b_canc[K // 2:, :] = b[: K // 2, :] * np.float32(1.0 + 2.0**-12)
    cases = {"random": (a, b, np.ones(R), np.ones(N)), "wide": (a_wide, b_wide, row, col), "cancelling": (a_canc, b_canc, np.ones(R), np.ones(N))}
    fn = HipContractor(tree)
    errs = {}
    for label, (x, y, rs, cs) in cases.items():
        ref = x.astype("complex128") @ y.astype("complex128")
        # error relative to the size of the terms: |a_i| . |b_j| per element
        terms = np.abs(x.astype("complex128")) @ np.abs(y.astype("complex128"))
        for mode in ("1", "0"):
            monkeypatch.setenv("CTG_PAIR_BF16X3", mode)
            got = np.asarray(fn(x, y)).astype("complex128")
            names = fn.setup(x, y)["exec"].step_kernels()
            assert any(n.startswith("pair_mfma_bf3_kernel" if (mode == "1" and eligible) else "pair_mfma_fast_kernel") for n in names), names
            errs[(label, mode)] = float((np.abs(got - ref) / terms).max())
    fn.close()
    print({k: f"{v:.2e}" for k, v in errs.items()})
    for label in cases:
        assert errs[(label, "0")] <= 2e-6
        assert errs[(label, "1")] <= max(1.5 * errs[(label, "0")], 2e-7), label


@pytest.mark.parametrize("name", ["C5_hyper200", "lattice8x8_sliced"])
def test_profile_slice_with_batched_launches_on_a_grouped_tree(name, monkeypatch):
    """``ctg_exec_profile_slice`` under ``CTG_PROFILE_SLICES`` (the per-step table of ``tools/steps_batched.py``)
    on an executor that batches whole slice groups: the profiled batch is whole groups (z = group * d + member,
    shared steps once per group) and what it adds to the result is exactly those slices."""
    from cotengra_amd import plan as P
    from cotengra_amd.distributed import scatter_slices

    monkeypatch.delenv("CTG_SLICE_GROUPS", raising=False)
    monkeypatch.setattr(P, "GROUP_MIN_WIDTH", 1)
    monkeypatch.setattr(P, "GROUP_MIN_SAVING", 0.0)
    monkeypatch.setattr(P, "GROUP_MIN_SAVING_SMALL", 0.0)
    case = next(c for c in G.cases("tree") if c["name"] == name)
    tree = G.tree_of(case)
    arrays = G.arrays_of(case, "complex128", tree)
    fn = HipContractor(tree)
    st = fn.setup(*arrays)
    ex, plan = st["exec"], st["plan"]
    gs = int(plan.group_size)
    if gs < 2 or ex.batch < 2 * gs:
        fn.close()
        pytest.skip("no batched slice groups on this tree")
    nb = 2 * gs
    monkeypatch.setenv("CTG_PROFILE_SLICES", str(nb))
    ex.zero_result()
    ms = ex.profile_slice(0)
    assert len(ms) == len(plan.steps) and np.all(np.isfinite(ms)) and np.all(np.asarray(ms) >= 0)
    got = np.asarray(ex.download_result())
    ids = [int(i) for i in plan.rank_slice_ids(0, 1, 0, 2)]
    assert len(ids) == nb
    a128 = [np.asarray(a).astype("complex128") for a in arrays]
    want = scatter_slices(tree, ids, [np.asarray(orc.contract_slice(tree, a128, i)) for i in ids])
    assert got.shape == want.shape and np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
    fn.close()


def test_bench_legs_of_the_multi_rank_run_under_a_one_rank_launcher():
    """``bench.py`` as the driver starts it at N > 1 (``torch.distributed.run``, RCCL process group, the
    collective behind the C ABI) with ONE rank and ``CTG_BENCH_C3_AMPLITUDES=1``: the legs that otherwise only
    exist at N > 1 -- m10 amplitudes per second and BASELINE config 3 as worded (one amplitude's 64 slices
    dealt over the ranks + reduce) -- run on the one-GPU lease and land in the compact line."""
    import json
    import subprocess
    import sys

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CTG_BENCH_C3_AMPLITUDES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = r.stdout.strip().splitlines()[-1]
    assert len(line) < 4096
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["roofline"]["frac"] > 0
    legs = rec["legs"]
    assert legs["C3_amplitudes_per_sec"] > 50 and legs["C3_strong_ms"] > 0
    full = json.load(open(os.path.join(ROOT, rec["full_record"])))
    strong = full["configs"]["C3_strong"]
    assert strong["nslices"] == 64 and strong["amplitude_rel_diff"] <= 1e-6
    assert full["config"]["reduce_via"] == "ctg_exec_reduce (RCCL, C ABI)"
