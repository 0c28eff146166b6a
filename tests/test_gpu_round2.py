"""GPU suite (round 2): the pieces added with C ABI v2, on the HIP path.

* two PROCESSES running the HIP executor, each with
  ``ctg_exec_run_slices(first=rank, stride=2)`` (reference ``contract_mpi``
  round-robin, core.py:4068-4076).  The box has one GPU and RCCL refuses two
  ranks on one device, so the two ranks share GPU 0 and exchange their
  downloaded partials over gloo; the per-rank executor, the device-side
  accumulation and the exponent-aware merge are the product code;
* the RCCL collective behind the C ABI (``ctg_comm_*`` / ``ctg_exec_reduce``)
  with a one-rank communicator, all-reduce and rooted, with and without
  ``strip_exponent``; an mpi4py-shaped communicator as the reference takes;
* checkpoint / resume: bit-identical to an uninterrupted run;
* the per-op plug-in with this package's own (einsum, tensordot);
* stream following, result ownership, projected output indices.
"""
import os
import socket
import sys

import numpy as np
import pytest

import cotengra_amd as ca
from cotengra_amd import runtime
from cotengra_amd.contractor import HipContractor
from oracle import contract_ref as orc

import golden_util as G

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TREE_CASES = G.cases("tree")
R2_CASES, R2_EXPECTED = G.load_r2()


def case_named(name):
    return next(c for c in TREE_CASES if c["name"] == name)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


# ---------------------------------------------------------------------- #
# two ranks, HIP executor in each
# ---------------------------------------------------------------------- #


def _rank_main(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import golden_util as G2
    from cotengra_amd import runtime as rt

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok, notes = True, []
    try:
        calls = []
        real = rt.Executor.run_share

        def spy(self, rank_=0, world_=1, unit_first=0, unit_count=-1):
            # (round 5: a rank's share goes through ctg_exec_run_share -- whole slice groups rank, rank + world, ...;
            # single slices round-robin, core.py:4070, for a plan without groups)
            units, per = self.plan.share_units(rank_, world_)
            calls.append((rank_, world_, unit_first, unit_count, units * per))
            return real(self, rank_, world_, unit_first, unit_count)

        rt.Executor.run_share = spy
        cases = {c["name"]: c for c in G2.cases("tree")}
        # inner-sliced lattice: numpy inputs and device inputs, all-reduce and rooted
        c = cases["lattice8x8_sliced"]
        tree = G2.tree_of(c)
        arrays = G2.arrays_of(c, "complex128", tree)
        ref = G2.expected("lattice8x8_sliced/complex128")
        out = tree.contract_distributed(arrays)
        ok &= G2.relerr(out, ref) < 1e-10
        mine = len(range(rank, tree.nslices, world))
        ok &= calls[-1][:3] == (rank, world, 0) and abs(calls[-1][4] - mine) <= 8   # (whole groups: within one group of the round-robin count)
        dev = [torch.as_tensor(a, device="cuda") for a in arrays]
        out = tree.contract_mpi(dev, root=1)
        ok &= (out is None) if rank != 1 else (out.is_cuda and G2.relerr(out.cpu().numpy(), ref) < 1e-10)
        # exponent-aware merge across ranks (un-rescaled lattice: value ~ 1e-34)
        m, e = tree.contract_distributed(arrays, strip_exponent=True)
        ok &= abs(complex(m) * 10.0**e - complex(ref)) <= 1e-10 * abs(complex(ref))
        xs64 = [a.astype("complex64") for a in arrays]
        m, e = tree.contract_distributed(xs64, strip_exponent=True)
        nm, ne = orc.contract(tree, xs64, strip_exponent=True)  # numpy in single precision
        tol64 = max(1e-5, 8.0 * abs(complex(nm) * 10.0**ne - complex(ref)) / abs(complex(ref)))
        ok &= abs(complex(m) * 10.0**e - complex(ref)) <= tol64 * abs(complex(ref))
        # output-sliced hyper network: chunks scatter-added on the device
        c2 = cases["rand_s42_r2_o2_hi1_ho2_outsliced"]
        t2 = G2.tree_of(c2)
        a2 = G2.arrays_of(c2, "complex128", t2)
        out2 = t2.contract_distributed(a2)
        ok &= G2.relerr(out2, G2.expected("rand_s42_r2_o2_hi1_ho2_outsliced/complex128")) < 1e-10
        # fewer slices than ranks: the reference's error (core.py:4062-4066)
        try:
            tree.unslice_all().contract_distributed(arrays)
            ok = False
        except ValueError:
            pass
        notes.append(calls[:1])
    except Exception as exc:  # report instead of hanging the peer
        ok = False
        notes.append(repr(exc))
    finally:
        q.put((rank, bool(ok), notes))
        dist.destroy_process_group()


def test_two_ranks_with_the_hip_executor():
    """Two processes, each with its own HIP executor on GPU 0, exchanging over gloo.  Not
    over RCCL, because RCCL refuses two ranks on one device -- measured on the lease
    (tools/exp_rccl_same_gpu.py, gpurun_out/r2a/rccl_same_gpu.log), both ranks:

        CommError: ncclCommInitRank(rank 1 of 2, device 0) failed: invalid usage
                   (run with NCCL_DEBUG=WARN for details)

    so on a one-GPU box the RCCL path of the C ABI is exercised with world = 1
    (test_cabi_collective_single_rank, tests/cabi_reduce.c) and the two-rank logic --
    round-robin shares, device-side accumulation, exponent merge -- here."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    assert [(r, ok) for r, ok, _ in results] == [(0, True), (1, True)], results
    assert all(p.exitcode == 0 for p in procs)


# ---------------------------------------------------------------------- #
# RCCL behind the C ABI
# ---------------------------------------------------------------------- #


def test_cabi_collective_single_rank():
    comm = runtime.Comm(runtime.Comm.unique_id(), 0, 1, device=0)
    try:
        c = case_named("lattice8x8_sliced")
        tree = G.tree_of(c)
        ref = G.expected("lattice8x8_sliced/complex128")
        for dtype, tol in (("complex128", 1e-10), ("complex64", None)):
            arrays = [a.astype(dtype) for a in G.arrays_of(c, "complex128", tree)]
            if tol is None:  # single precision: max(1e-5, 8 x numpy's own single-precision error)
                nm, ne = orc.contract(tree, arrays, strip_exponent=True)
                tol = max(1e-5, 8.0 * abs(complex(nm) * 10.0**ne - complex(ref)) / abs(complex(ref)))
            for root in (None, 0):
                # the package's driver with an explicit communicator
                m, e = tree.contract_mpi(arrays, comm=comm, root=root, strip_exponent=True)
                assert abs(complex(m) * 10.0**e - complex(ref)) <= tol * abs(complex(ref))
        # rescaled inputs: plain sum, complex64 within the single-precision gate
        c2 = case_named("lattice4x4_sliced")
        t2 = G.tree_of(c2)
        a2 = G.arrays_of(c2, "complex128", t2)
        out = t2.contract_distributed(a2, comm=comm)
        assert G.relerr(out, G.expected("lattice4x4_sliced/complex128")) < 1e-10
        # raw sequence a C caller would use: run, reduce in place, download
        fn = HipContractor(t2)
        st = fn.setup(*a2)
        ex = st["exec"]
        ex.zero_result()
        ex.run_slices(0, t2.nslices, 1)
        ex.reduce(comm, None)
        ex.reduce(comm, 0)  # a sum over one rank is the identity, twice as well
        assert G.relerr(ex.download_result(), G.expected("lattice4x4_sliced/complex128")) < 1e-10
        fn.close()
        with pytest.raises(ValueError):
            ex2 = HipContractor(t2)
            s2 = ex2.setup(*a2)
            try:
                s2["exec"].reduce(comm, 3)  # root outside the world
            finally:
                ex2.close()
    finally:
        comm.close()


class _OneRankMpi:
    """The three calls ``contract_mpi`` needs from an mpi4py communicator."""

    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1

    def bcast(self, obj, root=0):
        return obj


def test_contract_mpi_with_an_mpi_shaped_communicator():
    from cotengra_amd.distributed import close_comms

    c = case_named("lattice8x8_sliced")
    tree = G.tree_of(c)
    arrays = G.arrays_of(c, "complex128", tree)
    world = _OneRankMpi()
    try:
        out = tree.contract_mpi(arrays, comm=world)
        assert G.relerr(out, G.expected("lattice8x8_sliced/complex128")) < 1e-10
        assert G.relerr(tree.contract_mpi(arrays, comm=world, root=0), G.expected("lattice8x8_sliced/complex128")) < 1e-10
    finally:
        close_comms()


# ---------------------------------------------------------------------- #
# checkpoint / resume
# ---------------------------------------------------------------------- #


@pytest.mark.parametrize("strip", [False, True])
@pytest.mark.parametrize("name", ["lattice8x8_sliced", "rand_s42_r2_o2_hi1_ho2_outsliced"])
def test_resume_is_bit_identical(tmp_path, name, strip):
    c = case_named(name)
    tree = G.tree_of(c)
    assert tree.nslices >= 4
    arrays = G.arrays_of(c, "complex128", tree)
    whole = tree.contract(arrays, strip_exponent=strip)
    ck = str(tmp_path / "amp.ckpt")
    k = tree.nslices // 2 + 1
    assert tree.contract_resumable(arrays, ck, every=1, strip_exponent=strip, stop_after=k) is None
    assert os.path.exists(ck)
    # "the process died": forget every executor, then pick the run up from the file
    for fn in tree.contraction_cores.values():
        fn.close()
    tree.contraction_cores.clear()
    fresh = G.tree_of(c)
    out = fresh.contract_resumable(arrays, ck, every=2, strip_exponent=strip)
    assert not os.path.exists(ck)
    if strip:
        assert out[1] == whole[1]
        assert np.array_equal(np.asarray(out[0]), np.asarray(whole[0]))
    else:
        assert np.array_equal(np.asarray(out), np.asarray(whole))
    ref = G.expected(f"{name}/complex128")
    val = np.asarray(out[0]) * 10.0 ** out[1] if strip else np.asarray(out)
    assert G.relerr(val, ref) < 1e-10
    # a checkpoint of another contraction is refused
    assert fresh.contract_resumable(arrays, ck, every=1, stop_after=1) is None
    other = fresh.restore_ind(next(iter(fresh.sliced_inds)))
    with pytest.raises(ValueError):
        other.contract_resumable(arrays, ck)


def test_raw_state_round_trip_through_the_cabi():
    c = case_named("lattice4x4_sliced")
    tree = G.tree_of(c)
    arrays = G.arrays_of(c, "complex128", tree)
    a, b = HipContractor(tree), HipContractor(tree)
    sa, sb = a.setup(*arrays), b.setup(*arrays)
    sa["exec"].zero_result()
    sa["exec"].run_slices(0, 2, 1)
    part, e, z = sa["exec"].get_state()
    assert e == 0.0 and not z
    sb["exec"].set_state(part, e, z)
    sb["exec"].run_slices(2, tree.nslices - 2, 1)
    sa["exec"].run_slices(2, tree.nslices - 2, 1)
    assert np.array_equal(sa["exec"].download_result(), sb["exec"].download_result())
    with pytest.raises(ValueError):
        sb["exec"].set_state(np.zeros(7, dtype=np.complex128))
    a.close(), b.close()


# ---------------------------------------------------------------------- #
# per-op plug-in, streams, ownership, projections
# ---------------------------------------------------------------------- #


def test_per_op_plugin_on_the_gpu():
    import torch

    ca.interface.clear_expression_cache()
    c = case_named("lattice4x4_sliced")
    tree = G.tree_of(c)
    arrays = G.arrays_of(c, "complex128", tree)
    ref = G.expected("lattice4x4_sliced/complex128")
    got = tree.contract(arrays, implementation=(ca.einsum, ca.tensordot))
    assert G.relerr(got, ref) < 1e-10
    n_cached = len(ca.interface._EXPR_CACHE)
    assert 0 < n_cached <= ca.interface._EXPR_CACHE_SIZE
    got = tree.contract(arrays, implementation=(ca.einsum, ca.tensordot))  # plans reused
    assert len(ca.interface._EXPR_CACHE) == n_cached and G.relerr(got, ref) < 1e-10
    dev = [torch.as_tensor(a, device="cuda") for a in arrays]
    got = tree.contract(dev, implementation=(ca.einsum, ca.tensordot))
    assert got.is_cuda and G.relerr(got.cpu().numpy(), ref) < 1e-10
    # hyper network: einsum steps with batch indices, plus preprocessing
    c5 = case_named("rand_s42_r2_o2_hi1_ho2_outsliced")
    t5 = G.tree_of(c5)
    a5 = G.arrays_of(c5, "complex128", t5)
    got = t5.contract(a5, implementation=(ca.einsum, ca.tensordot))
    assert G.relerr(got, G.expected("rand_s42_r2_o2_hi1_ho2_outsliced/complex128")) < 1e-10
    ca.interface.clear_expression_cache()
    assert not ca.interface._EXPR_CACHE


def test_executor_follows_the_current_torch_stream():
    import torch

    c = case_named("lattice4x4_sliced")
    tree = G.tree_of(c)
    arrays = G.arrays_of(c, "complex128", tree)
    ref = G.expected("lattice4x4_sliced/complex128")
    dev = [torch.as_tensor(a, device="cuda") for a in arrays]
    out0 = tree.contract(dev)  # executor created on the default stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        # inputs produced on the side stream right before the call: the executor must
        # move to it, or its copies could run before these kernels
        scaled = [t * 2.0 for t in dev[:1]] + dev[1:]
        out1 = tree.contract(scaled)
        val1 = out1.cpu().numpy()
    side.synchronize()
    assert G.relerr(out0.cpu().numpy(), ref) < 1e-10
    assert G.relerr(val1, 2.0 * ref) < 1e-10
    out2 = tree.contract(dev)  # and back
    assert G.relerr(out2.cpu().numpy(), ref) < 1e-10


def test_returned_tensors_are_not_overwritten_by_later_calls():
    import socket as _s

    import torch
    import torch.distributed as dist

    c = case_named("lattice4x4_sliced")
    tree = G.tree_of(c)
    arrays = G.arrays_of(c, "complex128", tree)
    ref = G.expected("lattice4x4_sliced/complex128")
    dev = [torch.as_tensor(a, device="cuda") for a in arrays]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        first = tree.contract_distributed(dev)
        tree.contract_distributed([dev[0] * 3.0] + dev[1:])
        tree.contract([dev[0] * 5.0] + dev[1:])
        assert G.relerr(first.cpu().numpy(), ref) < 1e-10
    finally:
        from cotengra_amd.distributed import close_comms

        close_comms()
        dist.destroy_process_group()


@pytest.mark.parametrize("case", R2_CASES, ids=[c["name"] for c in R2_CASES])
def test_projected_output_index_on_the_gpu(case):
    tree = G.tree_of(case)
    for dtype, tol in (("complex128", 1e-10), ("complex64", 1e-5)):
        arrays = [a.astype(dtype) for a in G.arrays_of(case, "complex128", tree)]
        ref = R2_EXPECTED[f"{case['name']}/complex128"]
        got = np.asarray(tree.contract(arrays))
        assert got.shape == ref.shape
        assert G.relerr(got, ref) <= tol
        for i in case["slice_ids"]:
            sl = R2_EXPECTED[f"{case['name']}/complex128/slice{i}"]
            assert G.relerr(np.asarray(tree.contract_slice(arrays, i)), sl) <= tol
        chunks = list(tree.gen_output_chunks(arrays))
        assert sum(np.abs(np.asarray(ch)).sum() for ch in chunks) > 0


# ---------------------------------------------------------------------- #
# slice batching
# ---------------------------------------------------------------------- #


@pytest.mark.parametrize("dtype", ["complex64", "complex128"])
def test_slice_batching_does_not_change_a_bit(monkeypatch, dtype):
    """Up to 64 slices of a run share every launch (gridDim.y, one arena replica
    each).  The split of a contraction, the kernels and the order in which
    slices are added are those of one launch sequence per slice, so the result
    must be identical bit for bit, for any batch size."""
    import json

    names = ["lattice8x8_sliced", "rand_s42_r2_o2_hi1_ho2_outsliced", "lattice4x4_sliced"]
    for name in names:
        c = case_named(name)
        outs = []
        for cap in ("64", "3", "1"):
            monkeypatch.setenv("CTG_SLICE_BATCH", cap)
            tree = G.tree_of(c)
            arrays = [a.astype(dtype) for a in G.arrays_of(c, "complex128", tree)]
            outs.append(np.asarray(tree.contract(arrays)))
            for fn in tree.contraction_cores.values():
                fn.close()
        assert np.array_equal(outs[0], outs[2]) and np.array_equal(outs[1], outs[2]), name
        if dtype == "complex128" and c.get("rescale", False):
            assert G.relerr(outs[0], G.expected(f"{name}/complex128")) < 1e-10
    # the Sycamore m10 amplitude: 64 slices of 170 steps, streaming kernels included
    rec = ca.load_network(os.path.join(ROOT, "tests", "golden", "trees", "sycamore_m10.json"))
    z = np.load(os.path.join(ROOT, "tests", "golden", "sycamore_m10_arrays.npz"))
    amps = []
    for cap in ("64", "5", "1"):
        monkeypatch.setenv("CTG_SLICE_BATCH", cap)
        tree = ca.tree_from_record(rec)
        xs = [z[f"t{i}"].astype(dtype) for i in range(tree.N)]
        amps.append(complex(np.asarray(tree.contract(xs))))
        # a strided share of the slices, as one rank of a multi-GPU run takes it
        fn = HipContractor(tree)
        st = fn.setup(*xs)
        st["exec"].zero_result()
        st["exec"].run_slices(1, 21, 3)
        amps.append(complex(st["exec"].download_result()))
        fn.close()
        for f in tree.contraction_cores.values():
            f.close()
    assert amps[0] == amps[2] == amps[4] and amps[1] == amps[3] == amps[5]
    ref = complex(np.load(os.path.join(ROOT, "tests", "golden", "sycamore_m10_expected.npz"))["amplitude"])
    assert abs(amps[0] - ref) <= (1e-10 if dtype == "complex128" else 1e-5) * abs(ref)


# ---------------------------------------------------------------------- #
# wave-front groups
# ---------------------------------------------------------------------- #


@pytest.mark.parametrize("dtype", ["float32", "float64", "complex64", "complex128"])
def test_grouped_launches_do_not_change_a_bit(monkeypatch, dtype):
    """Small trees are emitted level by level and consecutive independent small
    steps share one launch.  Every output element is still computed by the same
    thread in the same order: identical bits with grouping on and off, with the
    reference's depth-first order, and under slice batching; fewer launches."""
    # (round 6: the LDS-resident subtrees replace most of these launches and have their own bit-identity test,
    # tests/test_gpu_round6.py; this one is about the wave-front groups of the ordinary steps)
    monkeypatch.setenv("CTG_NO_LDS_RUNS", "1")
    for name in ["C2_lattice8x8_d4", "C5_hyper200", "lattice8x8_sliced", "rand_s42_r2_o2_hi1_ho2_outsliced"]:
        c = case_named(name)
        outs, counts = [], []
        for groups, order in ((True, None), (False, None), (True, "dfs")):
            if groups:
                monkeypatch.delenv("CTG_NO_GROUPS", raising=False)
            else:
                monkeypatch.setenv("CTG_NO_GROUPS", "1")
            tree = G.tree_of(c)
            arrays = [np.real(a).astype(dtype) if not dtype.startswith("complex") else a.astype(dtype)
                      for a in G.arrays_of(c, "complex128", tree)]
            fn = HipContractor(tree, order=order)
            st = fn.setup(*arrays)
            ex = st["exec"]
            counts.append(ex.launch_count())
            ex.zero_result()
            ex.run_slices(0, min(tree.nslices, 70), 1)
            outs.append(np.array(ex.download_result()))
            fn.close()
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), name
        assert np.all(np.isfinite(outs[0]))
        (s0, l0), (s1, l1), _ = counts
        assert s0 == s1 == l1 and l0 <= s0
        if name.startswith("C"):
            assert l0 * 2 <= s0, (name, counts)


@pytest.mark.parametrize(
    "env",
    [{"CTG_GRAPH": "1", "CTG_SLICE_BATCH": "1"}, {"CTG_NO_LANE_TABLES": "1"}, {"CTG_NO_FAST_GROUPS": "1"}],
    ids=["graph-replay", "no-lane-tables", "no-fast-groups"],
)
def test_development_switches_keep_the_bits(monkeypatch, env):
    """The development switches of INTEGRATION.md select other launch mechanics
    (captured slice graph, per-block lane constants, fewer shared launches),
    never other arithmetic: identical results."""
    for name in ["lattice8x8_sliced", "C2_lattice8x8_d4"]:
        c = case_named(name)
        outs = []
        for on in (False, True):
            for k, v in env.items():
                if on:
                    monkeypatch.setenv(k, v)
                else:
                    monkeypatch.delenv(k, raising=False)
            tree = G.tree_of(c)
            arrays = [a.astype("complex64") for a in G.arrays_of(c, "complex128", tree)]
            fn = HipContractor(tree)
            st = fn.setup(*arrays)
            ex = st["exec"]
            ex.zero_result()
            ex.run_slices(0, tree.nslices, 1)
            ex.run_slices(0, tree.nslices, 1)   # (a second pass: the graph is replayed, not captured)
            outs.append(np.array(ex.download_result()))
            fn.close()
        assert np.array_equal(outs[0], outs[1]), (name, env)
        assert np.all(np.isfinite(outs[0])) and np.any(outs[0] != 0)
