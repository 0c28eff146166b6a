"""Slice-parallel contraction over the GPUs of one node.

The reference's only distributed execution is ``ContractionTree.contract_mpi``
(``cotengra/core.py:4032-4090``): ranks take slices round-robin
(``range(rank, nslices, size)``, :4070), sum them locally, and meet in ONE
``Allreduce`` / ``Reduce`` of the output tensor (:4081, :4089).  Here the
ranks are one process per MI355X, the local loop and sum run on the device
inside ``ctg_exec_run_share(rank, world)``, and the single
collective is RCCL over xGMI on the executor's own stream
(``ctg_exec_reduce``, ``include/ctg_hip.h``), reducing the resident result
tensor in place.  There is no other inter-GPU traffic: inputs are tiny and
replicated, the plan is identical.

Who the ranks are can be said in three ways (``comm=``):

* nothing / a ``torch.distributed`` process group -- the group only hands the
  128-byte RCCL id around when its backend is "nccl"; with backend "gloo"
  (CPU test rigs, or several ranks sharing one GPU, which RCCL refuses) the
  partials are downloaded and summed by gloo instead;
* a ``cotengra_amd.runtime.Comm`` made by the caller;
* an mpi4py-style communicator (``Get_rank / Get_size / bcast``), as the
  reference takes: it too only carries the unique id.

``executor_factory`` exists so the world-size-2 CPU tests (gloo) can inject a
numpy executor; the product default is the HIP contractor.
"""

from __future__ import annotations

import os
import socket
import threading

import numpy as np

_COMMS = {}  # id(group) / id(mpi comm) -> runtime.Comm
# creating a communicator is a collective (every rank must do it exactly once, in the same
# order): two host threads resolving the same group must not both create one
_COMMS_LOCK = threading.RLock()


def slices_of_rank(nslices, rank, world, plan=None):
    """This rank's slices.  Without a plan (or with one that has no slice groups): the round-robin
    partition of ``contract_mpi`` (core.py:4070).  With slice groups: whole groups ``rank, rank +
    world, ...`` (``Plan.rank_slice_ids``) -- what the HIP executor runs in ``ctg_exec_run_share``:
    the steps a group shares are computed once per group and on one rank only."""
    if plan is not None and plan.group_size > 1:
        return [int(i) for i in plan.rank_slice_ids(rank, world)]
    return range(rank, nslices, world)


def device_identity(device=None):
    """A string naming the physical GPU behind this process's current device:
    host + PCI location when torch exposes it, else host + visible-device mask
    + ordinal."""
    import torch

    if device is None:
        device = torch.cuda.current_device()
    host = socket.gethostname()
    try:
        prop = torch.cuda.get_device_properties(device)
        uuid = getattr(prop, "uuid", None)
        if uuid is not None:
            return f"{host}/{uuid}"
        bus = [getattr(prop, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
        if all(b is not None for b in bus):
            return f"{host}/pci:{bus[0]}:{bus[1]}:{bus[2]}"
    except Exception:
        pass
    mask = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", "all"))
    return f"{host}/{mask}/{device}"


def assert_distinct_devices(group=None):
    """Every rank must own a different GPU (one process per GPU).  Catches the
    classic omission of ``torch.cuda.set_device(LOCAL_RANK)``, after which all
    ranks silently share GPU 0.  Returns the list of identities."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    names = [None] * world
    dist.all_gather_object(names, device_identity(), group=group)
    if len(set(names)) != world:
        raise RuntimeError(
            f"{world} ranks but only {len(set(names))} distinct GPUs {sorted(set(names))}: "
            "call torch.cuda.set_device(LOCAL_RANK) in every rank before contracting."
        )
    return names


def _is_mpi_like(comm):
    return all(hasattr(comm, a) for a in ("Get_rank", "Get_size", "bcast"))


def _resolve_comm(comm):
    """-> (rank, world, runtime.Comm or None, torch group or None)"""
    from . import runtime

    if isinstance(comm, runtime.Comm):
        return comm.rank, comm.world, comm, None
    if comm is not None and _is_mpi_like(comm):
        import torch

        rank, world = comm.Get_rank(), comm.Get_size()
        with _COMMS_LOCK:
            c = _COMMS.get(("mpi", id(comm)))
            if c is None:
                uid = comm.bcast(runtime.Comm.unique_id() if rank == 0 else None, root=0)
                c = _COMMS[("mpi", id(comm))] = runtime.Comm(uid, rank, world, torch.cuda.current_device())
        return rank, world, c, None
    import torch.distributed as dist

    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised.")
    group = comm
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if dist.get_backend(group) != "nccl":
        return rank, world, None, group
    key = ("torch", id(group) if group is not None else 0)
    with _COMMS_LOCK:
        c = _COMMS.get(key)
        if c is None:
            assert_distinct_devices(group)
            c = _COMMS[key] = runtime.Comm.from_torch_group(group)
    return rank, world, c, group


def close_comms():
    """Destroy the cached RCCL communicators (before
    ``dist.destroy_process_group()``)."""
    with _COMMS_LOCK:
        for c in _COMMS.values():
            c.close()
        _COMMS.clear()


def contract_distributed(
    tree, arrays, group=None, root=None, executor_factory=None, order=None, comm=None,
    strip_exponent=False, check_zero=False, progbar=False,
):
    """Contract all slices of ``tree`` across the ranks of ``comm`` / ``group``.

    Returns the full output on every rank (``root=None``, all-reduce) or only
    on ``root`` (others get ``None``), like ``contract_mpi``.  Unlike
    ``contract_mpi``, trees whose sliced indices appear in the output are
    accepted.  With ``strip_exponent`` the return value is ``(mantissa,
    exponent)`` (the reference offers that only through ``contract_slice``
    kwargs; the ranks' exponents are merged by ``ctg_exec_reduce``).
    """
    # Sliced *output* indices: the reference refuses them here (core.py:4051-4055,
    # it would need a gather + stack).  On the device every slice is scatter-added
    # into its chunk of the full result tensor (``accum_kernel``), so a rank's
    # partial is the full tensor with its own chunks filled and zeros elsewhere --
    # the same single sum-reduce completes it.
    rank, world, ccomm, tgroup = _resolve_comm(comm if comm is not None else group)
    if tree.multiplicity < world:
        # core.py:4062-4066
        raise ValueError(
            f"Need to have more slices than processes, but have "
            f"{tree.multiplicity} and {world} respectively."
        )
    if executor_factory is not None:
        # (the injected executor is dealt the same share as the HIP executor: the plan -- host-side
        # only -- says whether the tree has slice groups)
        from .contractor import _result_dtype, _tree_contractor

        # (host_plan: no native plan handle, so this path needs neither a GPU nor the HIP library)
        plan = _tree_contractor(tree, order).host_plan(_result_dtype(arrays))
        mine = slices_of_rank(tree.multiplicity, rank, world, plan=plan)
        return _reduce_host(
            _injected_partial(tree, arrays, mine, executor_factory), None, tgroup, rank, root
        )

    from .contractor import _is_torch, _tree_contractor

    fn = _tree_contractor(tree, order)
    with fn._lock:   # (one host thread per executor: upload -> run -> exchange -> fetch)
        st = fn.setup(*[_to_local_device(x) for x in arrays])
        ex = st["exec"]
        ex.set_strip_exponent(strip_exponent, check_zero)
        ex.zero_result()
        # (whole slice groups rank, rank + world, ...: ctg_exec_run_share)
        fn.run_share(ex, rank, world, progbar if rank == 0 else False)
        if ccomm is not None:
            # RCCL on the executor's stream, in place on the resident result
            ex.reduce(ccomm, root)
            if root is not None and rank != root:
                return None
            # (a copy: the executor's buffer is rewritten by the next call)
            return fn._finish(st, strip_exponent, check_zero)
        # host exchange (gloo): partial (+ exponent) downloaded, summed on the CPU
        part, exponent, _zero = ex.get_state()
    as_torch = any(_is_torch(x) and x.is_cuda for x in arrays)
    out = _reduce_host(part, exponent if strip_exponent else None, tgroup, rank, root)
    if out is None:
        return None
    if strip_exponent:
        out, exponent = out
    if as_torch:
        out = out.to(st["result"].device)
    elif out.ndim == 0:
        out = out.numpy()[()]
    else:
        out = out.numpy()
    return (out, exponent) if strip_exponent else out


def _injected_partial(tree, arrays, mine, executor_factory):
    partial = np.array(executor_factory(tree, arrays, mine), order="C")  # (keeps 0-d results 0-d)
    full_shape = tree.gathered_shape()
    if tuple(partial.shape) != full_shape:
        raise ValueError(
            f"executor returned shape {tuple(partial.shape)}, expected the full output {full_shape} "
            "(use scatter_slices for outer-sliced trees)."
        )
    return partial


def _reduce_host(partial, exponent, group, rank, root):
    """Sum host partials over a (gloo) group; with ``exponent`` the partials
    are mantissas scaled by 10^exponent and are first brought to the largest
    exponent (core.py:163-172).  Returns a torch CPU tensor (or a pair)."""
    import torch
    import torch.distributed as dist

    t = torch.as_tensor(np.array(partial, order="C"))  # a copy; 0-d stays 0-d
    if exponent is not None:
        e = torch.tensor([exponent], dtype=torch.float64)
        dist.all_reduce(e, op=dist.ReduceOp.MAX, group=group)
        emax = float(e.item())
        if emax != float("-inf"):
            t = t * (0.0 if exponent == float("-inf") else 10.0 ** (exponent - emax))
        exponent = emax
    # complex tensors are reduced as pairs of reals (portable across backends)
    buf = torch.view_as_real(t) if t.is_complex() else t
    if root is None:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(buf, dst=root, op=dist.ReduceOp.SUM, group=group)
        if rank != root:
            return None
    return (t, exponent) if exponent is not None else t


def scatter_slices(tree, slice_ids, slices):
    """Sum per-slice results into the full output tensor (numpy): inner sliced
    indices add up, outer (output) sliced indices select the chunk -- what
    ``gather_slices`` (core.py:3825-3882) does, for an arbitrary subset of the
    slices.  Host-side helper for executors injected into
    ``contract_distributed``; the HIP executor does this on the device."""
    from .contractor import _chunk_index

    shape = tree.gathered_shape()
    out = None
    for i, x in zip(slice_ids, slices):
        x = np.asarray(x)
        if out is None:
            out = np.zeros(shape, dtype=x.dtype)
        out[_chunk_index(tree, tree.slice_key(i))] += x
    if out is None:
        raise ValueError("no slices given")
    return out


def _to_local_device(x):
    """numpy inputs are uploaded by the executor; torch inputs must live on
    this rank's GPU."""
    if type(x).__module__.split(".")[0] == "torch" and not x.is_cuda:
        return x.numpy()
    return x
