"""Slice-parallel contraction over the GPUs of one node.

The reference's only distributed execution is ``ContractionTree.contract_mpi``
(``cotengra/core.py:4032-4090``): ranks take slices round-robin
(``range(rank, nslices, size)``, :4070), sum them locally, and meet in ONE
``Allreduce`` / ``Reduce`` of the output tensor (:4081, :4089).  Here the
ranks are one process per MI355X, the local loop and sum run on the device
inside ``ctg_exec_run_slices(first=rank, stride=world)``, and the single
collective is an RCCL (``torch.distributed`` backend "nccl") all-reduce or
reduce over xGMI of the resident result tensor.  There is no other
inter-GPU traffic: inputs are tiny and replicated, the plan is identical.

``executor_factory`` exists so the world-size-2 CPU tests (gloo) can inject a
numpy executor; the product default is the HIP contractor.
"""

from __future__ import annotations

import numpy as np


def slices_of_rank(nslices, rank, world):
    """Round-robin partition of ``contract_mpi`` (core.py:4070)."""
    return range(rank, nslices, world)


def contract_distributed(
    tree, arrays, group=None, root=None, executor_factory=None, order=None
):
    """Contract all slices of ``tree`` across the ranks of ``group``.

    Returns the full output on every rank (``root=None``, all-reduce) or only
    on ``root`` (others get ``None``), like ``contract_mpi``.  Unlike
    ``contract_mpi``, trees whose sliced indices appear in the output are
    accepted.
    """
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised.")
    # Sliced *output* indices: the reference refuses them here (core.py:4051-4055,
    # it would need a gather + stack).  On the device every slice is scatter-added
    # into its chunk of the full result tensor (``accum_kernel``), so a rank's
    # partial is the full tensor with its own chunks filled and zeros elsewhere --
    # the same single sum-reduce completes it.
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if tree.multiplicity < world:
        # core.py:4062-4066
        raise ValueError(
            f"Need to have more slices than processes, but have "
            f"{tree.multiplicity} and {world} respectively."
        )

    mine = slices_of_rank(tree.multiplicity, rank, world)

    if executor_factory is None:
        from .contractor import _tree_contractor

        fn = _tree_contractor(tree, order)
        st = fn.setup(*[_to_local_device(x) for x in arrays])
        ex = st["exec"]
        ex.set_strip_exponent(False)
        ex.zero_result()
        ex.run_slices(rank, len(mine), world)
        if "result" in st:
            partial = st["result"]
        else:
            partial = torch.as_tensor(ex.download_result())
    else:
        partial = torch.as_tensor(
            np.ascontiguousarray(executor_factory(tree, arrays, mine))
        )
        full_shape = tuple(tree.size_dict[ix] for ix in tree.output)
        if tuple(partial.shape) != full_shape:
            raise ValueError(
                f"executor returned shape {tuple(partial.shape)}, expected the full output {full_shape} "
                "(use scatter_slices for outer-sliced trees)."
            )

    partial = partial.contiguous()
    on_host = not partial.is_cuda
    if on_host and dist.get_backend(group) == "nccl":
        # RCCL reduces device memory only (numpy inputs were downloaded above)
        partial = partial.cuda()
    # complex tensors are reduced as pairs of reals (portable across backends)
    buf = torch.view_as_real(partial) if partial.is_complex() else partial
    if root is None:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(buf, dst=root, op=dist.ReduceOp.SUM, group=group)
        if rank != root:
            return None
    return partial.cpu() if on_host else partial


def scatter_slices(tree, slice_ids, slices):
    """Sum per-slice results into the full output tensor (numpy): inner sliced
    indices add up, outer (output) sliced indices select the chunk -- what
    ``gather_slices`` (core.py:3825-3882) does, for an arbitrary subset of the
    slices.  Host-side helper for executors injected into
    ``contract_distributed``; the HIP executor does this on the device."""
    shape = tuple(tree.size_dict[ix] for ix in tree.output)
    out = None
    for i, x in zip(slice_ids, slices):
        x = np.asarray(x)
        if out is None:
            out = np.zeros(shape, dtype=x.dtype)
        key = tree.slice_key(i)
        idx = tuple(key[ix] if ix in key else slice(None) for ix in tree.output)
        out[idx] += x
    if out is None:
        raise ValueError("no slices given")
    return out


def _to_local_device(x):
    """numpy inputs are uploaded by the executor; torch inputs must live on
    this rank's GPU."""
    if type(x).__module__.split(".")[0] == "torch" and not x.is_cuda:
        return x.numpy()
    return x
