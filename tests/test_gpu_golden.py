"""GPU parity proper: the HIP path (Python host -> C ABI -> gfx950 kernels)
against the golden vectors frozen from the real reference, plus the oracle on
the same seeded inputs where no golden exists.

Tolerances: double precision 1e-10 relative to max|ref| (bit-level agreement
is not defined for floating point; the reference's own gate is 5e-6,
tests/test_compute.py:113).  Single precision: max(1e-5, 8 x the error numpy
itself makes when the oracle runs the same tree in the same single precision)
-- 1e-5 is the north-star tolerance; a result whose exact value is a
cancelling sum cannot be asked to beat fp32 rounding noise by more than that
(the reference's own single-precision gate is 5e-3).
"""
import numpy as np
import pytest

from cotengra_amd.contractor import HipContractor
from oracle import contract_ref as orc

import golden_util as G

pytestmark = pytest.mark.gpu

TREE_CASES = G.cases("tree")
EQ_CASES = G.cases("eq")
TOL = {"float64": 1e-10, "complex128": 1e-10}
LOW = {"complex128": "complex64", "float64": "float32"}
SINGLE = ("float32", "complex64")
NORTH_STAR = 1e-5
UNDERFLOW = 1e-30  # max|ref| below this cannot be represented by an fp32 result


def single_gate(ref, numpy_single):
    """Tolerance of a single-precision result: see the module docstring.
    ``numpy_single`` = the oracle's result on the same inputs in the same
    single precision (None: no such run, the plain north-star gate)."""
    if numpy_single is None:
        return NORTH_STAR
    return max(NORTH_STAR, 8.0 * G.relerr(np.asarray(numpy_single), ref))


def check(got, ref, dtype, numpy_single=None):
    got = np.asarray(got.cpu()) if hasattr(got, "cpu") else np.asarray(got)
    tol = single_gate(ref, numpy_single) if dtype in SINGLE else TOL[dtype]
    assert G.relerr(got, ref) <= tol, (G.relerr(got, ref), tol, dtype)


def check_stripped(pair, ref, dtype, numpy_single=None):
    """(mantissa, exponent) of a strip_exponent run against ``ref``: how a
    single-precision run reports values below the fp32 range."""
    m, e = pair
    m = np.asarray(m.cpu()) if hasattr(m, "cpu") else np.asarray(m)
    wide = m.astype("complex128" if np.iscomplexobj(m) else "float64") * 10.0**e
    check(wide, ref, dtype, numpy_single)


@pytest.mark.parametrize("case", TREE_CASES, ids=[c["name"] for c in TREE_CASES])
def test_tree_cases(case):
    tree = G.tree_of(case)
    for dt in case["dtypes"]:
        arrays = G.arrays_of(case, dt, tree)
        for run_dt in (dt, LOW[dt]):
            xs = [a.astype(run_dt) for a in arrays]
            single = run_dt in SINGLE
            if case["slice_ids"]:
                for i in case["slice_ids"]:
                    if i >= 2**62:
                        continue  # beyond the int64 slice ids of the C ABI
                    ref = G.expected(f"{case['name']}/{dt}/slice{i}")
                    if single and np.abs(ref).max() < UNDERFLOW:
                        # below the fp32 range: the value is carried by the exponent
                        np_pair = orc.contract_slice(tree, xs, i, strip_exponent=True)
                        np_single = np.asarray(np_pair[0]).astype(dt) * 10.0 ** np_pair[1]
                        check_stripped(tree.contract_slice(xs, i, strip_exponent=True), ref, run_dt, np_single)
                        continue
                    np_single = orc.contract_slice(tree, xs, i) if single else None
                    check(tree.contract_slice(xs, i), ref, run_dt, np_single)
            else:
                ref = G.expected(f"{case['name']}/{dt}")
                if single and np.abs(ref).max() < UNDERFLOW:
                    np_pair = orc.contract(tree, xs, strip_exponent=True)
                    np_single = np.asarray(np_pair[0]).astype(dt) * 10.0 ** np_pair[1]
                    check_stripped(tree.contract(xs, strip_exponent=True), ref, run_dt, np_single)
                    continue
                np_single = orc.contract(tree, xs) if single else None
                check(tree.contract(xs), ref, run_dt, np_single)


@pytest.mark.parametrize("case", EQ_CASES, ids=[c["name"] for c in EQ_CASES])
def test_reference_test_equations(case):
    for dt in ("complex128", "float64"):
        tree, arrays = G.eq_tree_and_arrays(case, dt)
        for run_dt in (dt, LOW[dt]):
            xs = [a.astype(run_dt) for a in arrays]
            np_single = orc.contract(tree, xs) if run_dt in SINGLE else None
            check(tree.contract(xs), G.expected(f"{case['name']}/{dt}"), run_dt, np_single)


def test_projected_slices_sum_to_total():
    """sum_j remove_ind(ix, project=j) == total (tests/test_tree.py:315-335)."""
    parts = [c for c in TREE_CASES if c["name"].startswith("project_") and c["name"] != "project_total"]
    total = next(c for c in TREE_CASES if c["name"] == "project_total")
    acc = 0
    for c in parts:
        tree = G.tree_of(c)
        acc = acc + np.asarray(tree.contract(G.arrays_of(c, "complex128", tree)))
    check(acc, G.expected("project_total/complex128"), "complex128")


def test_torch_device_inputs_and_expression_api():
    import torch

    import cotengra_amd as ca

    case = next(c for c in TREE_CASES if c["name"] == "lattice4x4_sliced")
    tree = G.tree_of(case)
    arrays = G.arrays_of(case, "complex128", tree)
    dev = [torch.as_tensor(a, device="cuda") for a in arrays]
    out = tree.contract(dev)
    assert out.is_cuda
    check(out, G.expected("lattice4x4_sliced/complex128"), "complex128")
    expr = ca.array_contract_expression(tree.inputs, tree.output, tree.size_dict, optimize=tree)
    check(expr(*arrays), G.expected("lattice4x4_sliced/complex128"), "complex128")
    # contract_core takes already-sliced arrays (core.py:3724-3773)
    sl = tree.slice_arrays(arrays, 1)
    got = tree.contract_core(sl)
    check(got, orc.contract_slice(tree, arrays, 1), "complex128")
    # gen_output_chunks / gather_slices round trip
    slices = [tree.contract_slice(arrays, i) for i in range(tree.nslices)]
    check(tree.gather_slices(slices), G.expected("lattice4x4_sliced/complex128"), "complex128")


def test_strip_exponent_and_chunks():
    case = next(c for c in TREE_CASES if c["name"].endswith("_outsliced") and c["stats"]["nslices"] > 1)
    tree = G.tree_of(case)
    arrays = G.arrays_of(case, "complex128", tree)
    ref = G.expected(f"{case['name']}/complex128")
    m, e = tree.contract(arrays, strip_exponent=True)
    check(np.asarray(m) * 10.0**e, ref, "complex128")
    y = sum((np.abs(np.asarray(ch)) ** 2).sum() for ch in tree.gen_output_chunks(arrays))
    assert abs(y - (np.abs(ref) ** 2).sum()) <= 1e-9 * (np.abs(ref) ** 2).sum()


@pytest.mark.parametrize("dtype", ["complex128", "complex64", "float64"])
def test_device_side_exponent_stripping(dtype):
    """reference tests/test_compute.py:217-248: un-rescaled 8x8 lattice, whose
    value (~1e-34) is below the fp32 normal range product-wise, sliced and
    unsliced, (m, p) -> m * 10**p; plus the oracle's own (mantissa, exponent)."""
    case = next(c for c in TREE_CASES if c["name"] == "lattice8x8")
    for name in ("lattice8x8", "lattice8x8_sliced"):
        c = next(x for x in TREE_CASES if x["name"] == name)
        tree = G.tree_of(c)
        base = "float64" if dtype == "float64" else "complex128"
        arrays = [a.astype(dtype) for a in G.arrays_of(c, base, tree)]
        ref = G.expected(f"{name}/{base}")
        m, e = tree.contract(arrays, strip_exponent=True)
        assert np.isfinite(e)
        m = np.asarray(m)
        assert 0.5 < np.abs(m).max() <= 1.0 + 1e-5 or tree.nslices > 1
        got = m.astype("complex128" if "complex" in dtype else "float64") * 10.0**e
        tol = 1e-10
        if dtype == "complex64":
            nm, ne = orc.contract(tree, arrays, strip_exponent=True)  # numpy in single precision
            tol = max(NORTH_STAR, 8.0 * abs(complex(nm) * 10.0**ne - ref) / abs(ref))
        assert abs(got - ref) <= tol * abs(ref), (abs(got - ref) / abs(ref), tol)
        # the oracle's (mantissa, exponent) pair agrees as a number as well
        om, oe = orc.contract(tree, G.arrays_of(c, base, tree), strip_exponent=True)
        assert abs(om * 10.0**oe - ref) <= 1e-10 * abs(ref)


def test_check_zero():
    import cotengra_amd as ca

    tree = ca.ContractionTree.from_path(["ab", "bc", "cd"], "ad", dict(a=3, b=4, c=5, d=2), path=[(0, 1), (0, 1)])
    xs = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=0)
    xs[1] = np.zeros_like(xs[1])
    assert tree.contract(xs, strip_exponent=True, check_zero=True) == (0.0, float("-inf"))
    out = tree.contract(xs)
    assert np.all(np.asarray(out) == 0)


@pytest.mark.parametrize(
    "fixture",
    ["sycamore_m20_w30.json", "sycamore_m20_w32.json", "sycamore_m20_w32_c512.json", "sycamore_m20_w32_c128.json",
     "sycamore_m20_w32_time.json", "sycamore_m20_native.json", "sycamore_m20_fused.json",
     "sycamore_m20_w33_bf3.json", "sycamore_m20_w32_r4.json", "sycamore_m20_w32_g.json"],
)
def test_full_size_properties_m20(fixture):
    """Size-independent checks at full slice width (2^30 first-search tree;
    2^32 refined trees, c512 = the benchmark's): (1) slicing identity -- a slice of the
    tree equals the sum of the two slices obtained by slicing one more index;
    (2) linearity in one input."""
    import cotengra_amd as ca
    import os

    rec = ca.load_network(os.path.join(os.path.dirname(__file__), "golden", "trees", fixture))
    tree = ca.tree_from_record(rec)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=42, dtype="complex64", rescale=True)
    coarse = HipContractor(tree)
    full = np.asarray(coarse.contract_slice(arrays, 5))
    # linearity: scaling one input tensor scales the slice amplitude
    arrays2 = list(arrays)
    arrays2[7] = arrays[7] * np.complex64(0.5 - 2.0j)
    lin = np.asarray(coarse.contract_slice(arrays2, 5))
    coarse.close()
    # The gate is the sub-slice chain's (tests/test_gpu_fullwidth.py (ii)): a single-precision
    # slice amplitude is within g = max(1e-5, 8 x the error numpy's own single precision makes on
    # this tree, narrowed to the width the oracle can run) of the exact value; two single-precision
    # results of the same exact value are therefore within 2 g of each other.  (Round 3 and before: a
    # flat 2e-4.)
    from oracle import contract_ref as orc

    small = tree.slice(target_size=2**20)
    a128 = [a.astype("complex128") for a in arrays]
    ref = complex(orc.contract_slice(small, a128, 3))
    g = max(1e-5, 8.0 * abs(complex(orc.contract_slice(small, arrays, 3)) - ref) / abs(ref))
    print(fixture, "single-precision gate", g, "linearity", abs(lin - full * (0.5 - 2.0j)) / abs(full * (0.5 - 2.0j)))
    assert abs(lin - full * (0.5 - 2.0j)) <= 2.0 * g * abs(full) * abs(0.5 - 2.0j)
    # one more sliced index: fine slices 2*5, 2*5+1 ... in the finer tree's numbering
    big = max((p for p, _, _ in tree.traverse()), key=tree.get_size)
    ix = next(iter(tree.get_legs(big)))
    fine = tree.remove_ind(ix)
    key = tree.slice_key(5)
    ids = []
    for v in range(tree.size_dict[ix]):
        k = dict(key)
        k[ix] = v
        strides = ca.get_slice_strides(fine.sliced_inds)
        ids.append(sum(k[s] * st for s, st in zip(fine.sliced_inds, strides)))
    fc = HipContractor(fine)
    parts = sum(np.asarray(fc.contract_slice(arrays, i)) for i in ids)
    fc.close()
    print(fixture, "slicing identity", abs(parts - full) / abs(full))
    assert abs(parts - full) <= 2.0 * g * abs(full)


def test_contract_distributed_rccl_single_rank():
    """The slice-parallel driver on the GPU with the RCCL backend (one rank --
    the box has one GPU; ranks > 1 are covered by the gloo CPU test)."""
    import socket

    import torch
    import torch.distributed as dist

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    case = next(c for c in TREE_CASES if c["name"] == "lattice8x8_sliced")
    tree = G.tree_of(case)
    arrays = G.arrays_of(case, "complex128", tree)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        out = tree.contract_distributed([torch.as_tensor(a, device="cuda") for a in arrays])
        check(out, G.expected("lattice8x8_sliced/complex128"), "complex128")
        out0 = tree.contract_distributed(arrays, root=0)
        check(out0, G.expected("lattice8x8_sliced/complex128"), "complex128")
        # output-sliced tree: chunks scatter-added on the device, same single reduce
        case2 = next(c for c in TREE_CASES if c["name"] == "rand_s42_r2_o2_hi1_ho2_outsliced")
        tree2 = G.tree_of(case2)
        arrays2 = G.arrays_of(case2, "complex128", tree2)
        out2 = tree2.contract_distributed(arrays2)
        check(out2, G.expected("rand_s42_r2_o2_hi1_ho2_outsliced/complex128"), "complex128")
    finally:
        from cotengra_amd.distributed import close_comms

        close_comms()
        dist.destroy_process_group()


def test_benchmark_api():
    case = next(c for c in TREE_CASES if c["name"] == "lattice8x8_sliced")
    tree = G.tree_of(case)
    res = tree.benchmark(dtype="complex64", max_time=0.2, min_reps=3, max_reps=20)
    assert set(res) == {"time_per_slice", "est_time_total", "est_gigaflops"}
    assert res["time_per_slice"] > 0 and res["est_gigaflops"] > 0
    assert abs(res["est_time_total"] - res["time_per_slice"] * tree.nslices) < 1e-9
