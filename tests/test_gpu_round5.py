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
    ref = np.asarray(orc.contract(tree, arrays))
    fn = HipContractor(tree)
    st = fn.setup(*arrays)
    ex, plan = st["exec"], st["plan"]
    if groups and plan.group_size < 2:
        fn.close()
        pytest.skip("no step is independent of a sliced index")
    scale = np.abs(ref).max()
    for world in (1, 2, 3, 5):
        ex.zero_result()
        for rank in range(world):
            units, gs = plan.share_units(rank, world)
            if rank % 2:
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
    case = next(c for c in G.cases("tree") if c["name"] == "C5_hyper200")
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
        for name in ("rand_s42_r3_o1_hi0_ho0_outsliced", "C5_hyper200", "lattice8x8_sliced"):
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
