"""CPU suite, part 4: the slice-parallel multi-process path with the gloo
backend, world_size 2 (the GPU path uses the same driver with RCCL).

The per-rank executor is injected (numpy oracle summing the rank's slices) so
that the partitioning, the single collective, root/all-reduce semantics and
the reference's error behaviour (core.py:4051-4066) are exercised without a GPU.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, root, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import cotengra_amd as ca
    from cotengra_amd.distributed import contract_distributed, scatter_slices, slices_of_rank
    from oracle import contract_ref as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        inputs, output, shapes, size_dict = ca.lattice_equation([3, 4], d_min=2)
        inputs = [list(t) for t in inputs]
        inputs[0].append("Z")
        size_dict = dict(size_dict, Z=3)
        tree = ca.ContractionTree.from_path(inputs, ["Z"], size_dict,
                                            path=ca.greedy_path(inputs, ["Z"], size_dict))
        for ix in (inputs[5][0], inputs[6][1], inputs[2][0]):
            tree.remove_ind_(ix)
        arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=5, dtype="complex128")

        def executor(tree_, arrays_, mine):
            # (whole slice groups rank, rank + world, ... when the plan has groups; core.py:4070's
            # round-robin otherwise)
            from cotengra_amd import plan as P

            plan_ = P.compile_tree(tree_, "complex128")
            want = list(slices_of_rank(tree_.nslices, rank, world, plan=plan_))
            assert list(mine) == want
            if plan_.group_size == 1:
                assert want == list(range(rank, tree_.nslices, world))
            return sum(orc.contract_slice(tree_, arrays_, i) for i in mine)

        out = contract_distributed(tree, arrays, root=root, executor_factory=executor)
        ref = orc.contract(tree, arrays)
        ok = True
        if root is None or rank == root:
            ok = bool(np.allclose(out.numpy(), ref, rtol=1e-12, atol=1e-14))
        else:
            ok = out is None
        # outer-sliced output index: refused by contract_mpi (core.py:4051-4055),
        # supported here -- every rank scatters its chunks, one sum completes them
        t2 = tree.copy()
        t2.remove_ind_("Z")

        def executor2(tree_, arrays_, mine):
            return scatter_slices(tree_, mine, [orc.contract_slice(tree_, arrays_, i) for i in mine])

        out2 = contract_distributed(t2, arrays, root=root, executor_factory=executor2)
        if root is None or rank == root:
            ok = ok and bool(np.allclose(out2.numpy(), ref, rtol=1e-12, atol=1e-14))
        else:
            ok = ok and out2 is None
        # scalar output (an amplitude): the partial is a 0-d array all the way
        t4 = ca.ContractionTree.from_path(inputs, [], size_dict, path=ca.greedy_path(inputs, [], size_dict))
        t4.remove_ind_(inputs[5][0])
        out4 = contract_distributed(t4, arrays, root=root, executor_factory=executor)
        if root is None or rank == root:
            ok = ok and out4.shape == () and bool(np.allclose(out4.numpy(), orc.contract(t4, arrays), rtol=1e-12))
        # error behaviour mirrored from contract_mpi
        t3 = tree.unslice_all()
        try:
            contract_distributed(t3, arrays, executor_factory=executor)
            ok = False
        except ValueError:
            pass
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("root", [None, 0, 1])
def test_slice_parallel_gloo_world2(root):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, root, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
    results = sorted(q.get(timeout=10) for _ in procs)
    assert results == [(0, True), (1, True)]
    assert all(p.exitcode == 0 for p in procs)
