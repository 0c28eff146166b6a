"""The whole-tree HIP contractor and the execution drivers behind
``ContractionTree.contract / contract_core / contract_slice / gather_slices``.

The reference offers two plug-in points for execution:
per-op ``implementation=(einsum, tensordot)`` and the whole-tree backend
``CuQuantumContractor`` chosen in ``make_contractor``
(``cotengra/contract.py:840-1006``).  ``HipContractor`` fills the second slot
for MI355X: it is built once from a tree, lazily set up on the first call
(plan compilation + device residency), and then called with the arrays.

Arrays may be numpy arrays (copied host->device once per call) or torch
tensors already on a ROCm device (device-to-device copy into the executor's
input space, result returned as a torch tensor on the same device).
"""

from __future__ import annotations

import time

import threading

import numpy as np

from . import runtime
from .plan import DTYPE_CODES, compile_tree
from .utils import prod

_SUPPORTED = tuple(DTYPE_CODES)


class _Progress:
    """Slice counter of a long sliced run: a ``tqdm`` bar when tqdm is importable (the
    reference's ``progbar``, contract.py:785-790 / core.py:4010-4013), plain lines on
    stderr otherwise.  ``progbar`` may also be a callable ``f(done, total)``."""

    def __init__(self, progbar, total, desc="slices"):
        self.total, self.done, self.bar, self.call = int(total), 0, None, None
        if callable(progbar):
            self.call = progbar
        elif progbar:
            try:
                from tqdm import tqdm

                self.bar = tqdm(total=self.total, desc=desc, unit="slice")
            except ImportError:
                self.bar = False

    @property
    def active(self):
        return self.call is not None or self.bar is not None

    def update(self, n):
        self.done += n
        if self.call is not None:
            self.call(self.done, self.total)
        elif self.bar:
            self.bar.update(n)
        elif self.bar is False:
            import sys

            print(f"  {self.done} / {self.total} slices", file=sys.stderr, flush=True)

    def close(self):
        if self.bar:
            self.bar.close()


def inputs_digest(arrays):
    """sha256 over the shapes, dtypes and bytes of the input tensors (a sliced Sycamore
    network: 195 KB): what makes a checkpoint belong to THESE inputs, not just this tree."""
    import hashlib

    h = hashlib.sha256()
    for x in arrays:
        if _is_torch(x):
            x = x.detach().cpu().numpy()
        x = np.ascontiguousarray(np.asarray(x))
        h.update(repr((x.shape, str(x.dtype))).encode())
        h.update(x.tobytes())
    return h.hexdigest()


def _current_device():
    """The process's current ROCm device (one process per GPU sets it with
    ``torch.cuda.set_device(LOCAL_RANK)``); 0 without torch."""
    try:
        import torch

        if torch.cuda.is_available():
            return torch.cuda.current_device()
    except ImportError:
        pass
    return 0


def _release_execs(fn):
    """Close the idle executors of contractor ``fn``: not those another thread is inside
    of (its lock is held), nor those a suspended ``gen_output_chunks`` generator or a
    caller-visible state still uses (``pinned``).  True if anything was freed."""
    execs = getattr(fn, "_execs", None)
    lock = getattr(fn, "_lock", None)
    if not execs or lock is None or not lock.acquire(blocking=False):
        return False
    freed = False
    try:
        for key in [k for k, st in execs.items() if not st.get("pinned")]:
            st = execs.pop(key)
            st["exec"].close()
            st.pop("result", None)
            freed = True
    finally:
        lock.release()
    return freed


def _is_out_of_memory(exc):
    return isinstance(exc, MemoryError) or "out of memory" in str(exc).lower()


def _is_torch(x):
    return type(x).__module__.split(".")[0] == "torch"


def _result_dtype(arrays):
    names = []
    for x in arrays:
        dt = str(x.dtype).replace("torch.", "")
        names.append(dt)
    dt = np.result_type(*[np.dtype(n) for n in names]).name
    if dt not in _SUPPORTED:
        # integers / half precision are promoted like numpy would for matmul
        dt = np.result_type(np.dtype(dt), np.float32).name
        if dt not in _SUPPORTED:
            raise TypeError(f"Unsupported array dtype {dt}.")
    return dt


class HipContractor:
    """Callable performing the contraction of ``tree`` on an MI355X.

    Parameters
    ----------
    tree : ContractionTree
    order : str or callable, optional
        Traversal order (``ContractionTree.traverse``).
    strip_exponent, check_zero, progbar
        Defaults that a call may override, as for the reference's
        ``Contractor`` (contract.py:718-742).
    handle_slicing : bool
        True: calls take the *unsliced* arrays and return the full output
        (all slices run and are gathered on the device).  False: calls take
        arrays that were already sliced (``tree.contract_core`` semantics,
        core.py:3724-3773).
    device : int, optional
        GPU ordinal for numpy inputs (torch inputs bring their own device).
    """

    def __init__(
        self,
        tree,
        order=None,
        strip_exponent=False,
        check_zero=False,
        progbar=False,
        handle_slicing=True,
        device=None,
        force_kernel=None,
        fuse=None,
        fuse_min_elems=None,
        stem_bf16x3=None,
    ):
        self._origin = tree  # whose ``contraction_cores`` hold this contractor's siblings
        if handle_slicing or not tree.sliced_inds:
            self.tree = tree
        else:
            self.tree = _sliced_twin(tree)
        self.order = order
        self.strip_exponent = strip_exponent
        self.check_zero = check_zero
        self.progbar = progbar
        self.device = device
        self.force_kernel = force_kernel
        # fused stem pairs (plan.compile_tree): None = the default rule
        self.fuse = fuse
        self.fuse_min_elems = fuse_min_elems
        # arithmetic of the fused pairs: bf16 x 3 -- fp32 operands split exactly three ways, products on
        # the bf16 matrix cores -- unless switched off here (False) or by CTG_STEM_BF16X3=0
        # in the environment, which the kernel launcher reads at every launch and which wins
        # (round 6) a string names the arithmetic outright: "fp32", "bf16x3" (three exact limbs, six products) or
        # "fp16x2" (two limbs, three products: what an executor takes when nothing is said); True / False as
        # before: bf16 x 3 / fp32
        self.stem_arith = stem_bf16x3 if isinstance(stem_bf16x3, str) else None
        if isinstance(stem_bf16x3, str):
            stem_bf16x3 = stem_bf16x3 != "fp32"
        self.stem_bf16x3 = None if stem_bf16x3 is None else bool(stem_bf16x3)
        self._plans = {}  # dtype -> (Plan, DevicePlan)
        self._execs = {}  # (dtype, device, torch?) -> state dict
        # A ctg_exec is confined to one host thread at a time (include/ctg_hip.h); contractors
        # are cached on the tree and in the expression cache, and the reference's callers may
        # be multi-threaded (presets.py:77-88): upload -> run -> fetch is one critical section.
        self._lock = threading.RLock()

    # ------------------------------------------------------------------ #

    def host_plan(self, dtype):
        """The compiled plan alone -- pure Python, no native library: what a caller needs that only asks how
        the slices are dealt to ranks (``Plan.rank_slice_ids``)."""
        try:
            return self._plans[dtype][0]
        except KeyError:
            pass
        try:
            return self._host_plans[dtype]
        except (KeyError, AttributeError):
            if not hasattr(self, "_host_plans"):
                self._host_plans = {}
            plan = self._host_plans[dtype] = compile_tree(
                self.tree, dtype, order=self.order, force_kernel=self.force_kernel,
                fuse=self.fuse, fuse_min_elems=self.fuse_min_elems, stem_bf16x3=self.stem_bf16x3,
            )
            return plan

    def get_plan(self, dtype):
        try:
            return self._plans[dtype]
        except KeyError:
            plan = self.host_plan(dtype)
            entry = self._plans[dtype] = (plan, runtime.DevicePlan(plan))
            return entry

    def _get_exec(self, dtype, device, use_torch):
        key = (dtype, device, use_torch)
        try:
            return self._execs[key]
        except KeyError:
            pass
        plan, dplan = self.get_plan(dtype)
        try:
            st = self._new_exec(plan, dplan, dtype, device, use_torch)
        except (MemoryError, RuntimeError) as exc:
            # Every cached contractor of a tree (``contract`` and ``contract_core``,
            # each dtype) keeps its own arena resident -- tens of GiB for a wide tree.
            # When the device is full, give up the siblings' executors (they are
            # rebuilt on demand) and try once more.
            if not _is_out_of_memory(exc):
                raise
            # (also the one-shot expressions the einsum / tensordot front ends keep)
            from .interface import evict_expression_cache

            freed = self._evict_siblings()
            freed = evict_expression_cache(keep=self) or freed
            if not freed:
                raise
            st = self._new_exec(plan, dplan, dtype, device, use_torch)
        self._execs[key] = st
        return st

    def _evict_siblings(self):
        """Close the executors of the other contractors cached on the same tree
        (and this contractor's own executors for other dtypes / devices).
        Returns True if anything was freed."""
        freed = False
        cores = getattr(self._origin, "contraction_cores", {})
        for other in list(cores.values()) + [self]:
            freed = _release_execs(other) or freed
        if freed:
            try:
                import torch

                torch.cuda.empty_cache()  # result tensors went through torch's allocator
            except ImportError:
                pass
        return freed

    def _new_exec(self, plan, dplan, dtype, device, use_torch):
        st = {"plan": plan}
        if use_torch:
            import torch

            dev = torch.device("cuda", device)
            res = torch.zeros(
                plan.result_shape, dtype=getattr(torch, dtype), device=dev
            )
            stream = torch.cuda.current_stream(dev).cuda_stream
            st["result"] = res
            st["stream"] = stream
            st["exec"] = runtime.Executor(
                dplan, device=device, stream=stream, result_ptr=res.data_ptr()
            )
        else:
            st["exec"] = runtime.Executor(dplan, device=device)
        if self.stem_arith is not None:
            st["exec"].set_stem_arithmetic(self.stem_arith)
        elif self.stem_bf16x3 is not None:
            st["exec"].set_stem_arithmetic(self.stem_bf16x3)
        return st

    def setup(self, *arrays):
        """Make ``arrays`` resident and return the executor state (the
        analogue of ``CuQuantumContractor.setup``, contract.py:883-899)."""
        if len(arrays) != self.tree.N:
            raise ValueError(
                f"Expected {self.tree.N} arrays, got {len(arrays)}."
            )
        shapes = self.tree.get_shapes()
        for i, (x, s) in enumerate(zip(arrays, shapes)):
            if tuple(x.shape) != tuple(s):
                raise ValueError(
                    f"Array {i} has shape {tuple(x.shape)} but the tree "
                    f"expects {tuple(s)}."
                )
        dtype = _result_dtype(arrays)
        torch_in = [x for x in arrays if _is_torch(x) and x.is_cuda]
        if torch_in:
            import torch

            device = torch_in[0].device.index or 0
            st = self._get_exec(dtype, device, True)
            # the executor follows torch's *current* stream: uploads, kernels and the
            # caller's own tensor ops (input conversion, result clone) stay ordered
            cur = torch.cuda.current_stream(torch_in[0].device).cuda_stream
            if cur != st["stream"]:
                st["exec"].set_stream(cur)
                st["stream"] = cur
            tdt = getattr(torch, dtype)
            keep = []
            for x in arrays:
                if not _is_torch(x):
                    x = torch.as_tensor(np.asarray(x))
                keep.append(
                    x.to(device=torch_in[0].device, dtype=tdt).contiguous()
                )
            st["exec"].upload_device(
                [t.data_ptr() for t in keep], [t.numel() for t in keep]
            )
            st["keep"] = keep
        else:
            device = self.device if self.device is not None else _current_device()
            st = self._get_exec(dtype, device, False)
            host = [
                x.detach().cpu().numpy() if _is_torch(x) else np.asarray(x)
                for x in arrays
            ]
            st["exec"].upload_host(host)
        return st

    def _finish(self, st, strip_exponent, check_zero, index=None):
        """Fetch the result; with ``strip_exponent`` return ``(mantissa,
        exponent)`` as accumulated on the device (reference contract.py:834-835
        and core.py:125-172)."""
        if "result" in st:
            out = st["result"]
            out = out[index] if index is not None else out
            out = out.clone()
        else:
            out = st["exec"].download_result()
            if index is not None:
                out = out[index]
            if out.ndim == 0 and not strip_exponent:
                return out[()]
        if not strip_exponent:
            return out
        exponent, zero = st["exec"].get_exponent()
        if zero and check_zero and exponent == float("-inf"):
            return 0.0, float("-inf")
        if not _is_torch(out) and out.ndim == 0:
            out = out[()]
        return out, exponent

    def run_slices(self, ex, first, count, stride, progbar=False):
        """``count`` slices ``first, first + stride, ...``; with ``progbar`` in chunks (a
        multiple of the executor's slice batch, sized for a few updates per second) with a
        synchronisation in between, so that the counter shows slices that are DONE."""
        prog = _Progress(progbar, count) if progbar and count > 1 else None
        plan = ex.plan
        ids = None
        if plan.group_size > 1 and count > 1:
            # slice groups: the slices asked for, group by group (what a group shares is computed once
            # per group among them) -- in one call, or in chunks of whole groups under a progress bar
            ids = np.arange(first, first + count * stride, stride, dtype=np.int64)
            ids = ids[np.lexsort((ids, plan.group_of(ids)))]
        if prog is None or not prog.active:
            if ids is None:
                ex.run_slices(first, count, stride)
            else:
                ex.run_slice_list(ids)
            return
        import time

        chunk = max(int(ex.batch), 1) if ids is None else int(plan.group_size)
        done = 0
        try:
            while done < count:
                n = min(chunk, count - done)
                t0 = time.perf_counter()
                if ids is not None:
                    ex.run_slice_list(ids[done:done + n])
                else:
                    ex.run_slices(first + done * stride, n, stride)
                ex.sync()
                dt = time.perf_counter() - t0
                done += n
                prog.update(n)
                if dt < 0.1:   # tiny slices: fewer, larger chunks
                    chunk *= 2
        finally:
            prog.close()

    def run_share(self, ex, rank=0, world=1, progbar=False):
        """``rank``'s share of the slices (``Plan.share_units``: whole slice groups ``rank, rank + world,
        ...``; ``contract_mpi``'s round-robin, core.py:4070, for a plan without groups) through
        ``ctg_exec_run_share`` -- in one call, or under ``progbar`` in chunks of whole units with a
        synchronisation in between, so that the counter shows slices that are DONE."""
        units, gs = ex.plan.share_units(rank, world)
        prog = _Progress(progbar, units * gs) if progbar and units * gs > 1 else None
        if prog is None or not prog.active:
            ex.run_share(rank, world, 0, units)
            return
        import time

        chunk = max(int(ex.batch) // gs, 1)
        done = 0
        try:
            while done < units:
                n = min(chunk, units - done)
                t0 = time.perf_counter()
                ex.run_share(rank, world, done, n)
                ex.sync()
                dt = time.perf_counter() - t0
                done += n
                prog.update(n * gs)
                if dt < 0.1:   # tiny slices: fewer, larger chunks
                    chunk *= 2
        finally:
            prog.close()

    def __call__(self, *arrays, **kwargs):
        backend = kwargs.pop("backend", None)  # noqa: F841  (inferred from arrays)
        progbar = kwargs.pop("progbar", self.progbar)
        check_zero = kwargs.pop("check_zero", self.check_zero)
        strip_exponent = kwargs.pop("strip_exponent", self.strip_exponent)
        kwargs.pop("implementation", None)
        if kwargs:
            raise TypeError(f"Unknown keyword arguments: {kwargs}.")
        with self._lock:
            st = self.setup(*arrays)
            ex = st["exec"]
            ex.set_strip_exponent(strip_exponent, check_zero)
            ex.zero_result()
            self.run_share(ex, 0, 1, progbar)
            return self._finish(st, strip_exponent, check_zero)

    def contract_slice(self, arrays, i, strip_exponent=False, check_zero=False):
        """Output of slice ``i`` only (sliced output indices removed)."""
        with self._lock:
            st = self.setup(*arrays)
            ex = st["exec"]
            ex.set_strip_exponent(strip_exponent, check_zero)
            ex.zero_result()
            ex.run_slices(int(i), 1, 1)
            index = _chunk_index(self.tree, self.tree.slice_key(int(i)))
            return self._finish(st, strip_exponent, check_zero, index=index)

    def profile(self, arrays, slice_id=0):
        """Per-step milliseconds for one slice (see ``Plan.describe_steps``)."""
        with self._lock:
            st = self.setup(*arrays)
            return st["plan"], st["exec"].profile_slice(slice_id)

    def close(self):
        with self._lock:
            for st in self._execs.values():
                st["exec"].close()
            self._execs.clear()
            for _, dplan in self._plans.values():
                dplan.close()
            self._plans.clear()


class PerOpContractor:
    """The reference's per-op plug-in point: ``implementation=(einsum,
    tensordot)`` (``cotengra/contract.py:775-776`` -- in that order).  The tree
    is walked on the host and every pairwise step is handed to the caller's two
    functions:

    * ``tensordot(a, b, (axes_a, axes_b))`` for steps that are plain tensor
      products over shared indices, the result being ``[free-a..., free-b...]``
      and then transposed to the parent's index order (contract.py:808-812);
    * ``einsum(eq, a, b)`` for steps with batch / hyper indices, and
      ``einsum(eq, x)`` for single-tensor preprocessing (contract.py:795-800, 814).

    Passing this package's own pair, ``(cotengra_amd.einsum,
    cotengra_amd.tensordot)``, runs every step as its own small HIP plan --
    correct but one materialised tensor per step, which is what the whole-tree
    ``HipContractor`` exists to avoid (SURVEY section 8b calls this slot the
    fallback boundary).  Arrays are whatever the two functions accept; the only
    things asked of the arrays themselves are ``.transpose(*perm)`` / ``.permute``
    and, with ``strip_exponent``, ``abs(x).max()`` and division by a scalar.
    """

    def __init__(self, tree, implementation, order=None, prefer_einsum=False,
                 strip_exponent=False, check_zero=False):
        try:
            self._einsum, self._tensordot = implementation
        except (TypeError, ValueError):
            raise ValueError(
                "implementation must be 'hip' or a pair of callables (einsum, tensordot), "
                f"got {implementation!r}."
            ) from None
        if not (callable(self._einsum) and callable(self._tensordot)):
            raise ValueError("implementation=(einsum, tensordot) must be two callables.")
        self.strip_exponent = strip_exponent
        self.check_zero = check_zero
        self.n_inputs = tree.N
        # schedule: ("pair", out, left, right, axes | None, eq | None, perm | None)
        self.schedule = []
        if tree.N == 1:
            self.single = tree.get_eq_sliced()
            return
        self.single = None
        slot = {leaf: i for i, leaf in enumerate(tree.gen_leaves())}
        nxt = len(slot)
        for parent, left, right in tree.traverse(order=order):
            a, b = slot.pop(left), slot.pop(right)
            slot[parent] = nxt
            if not prefer_einsum and tree.get_can_dot(parent):
                self.schedule.append(
                    (nxt, a, b, tree.get_tensordot_axes(parent), None, tree.get_tensordot_perm(parent))
                )
            else:
                self.schedule.append((nxt, a, b, None, tree.get_einsum_eq(parent), None))
            nxt += 1
        self.root = nxt - 1
        # (filled lazily by the index queries above, so read afterwards)
        self.pre = dict(tree.preprocessing)

    @staticmethod
    def _transpose(x, perm):
        if hasattr(x, "permute"):
            return x.permute(*perm)
        return x.transpose(*perm)

    def __call__(self, *arrays, **kwargs):
        kwargs.pop("backend", None)
        kwargs.pop("progbar", None)
        strip = kwargs.pop("strip_exponent", self.strip_exponent)
        check_zero = kwargs.pop("check_zero", self.check_zero)
        if kwargs:
            raise TypeError(f"Unknown keyword arguments: {kwargs}.")
        if len(arrays) != self.n_inputs:
            raise ValueError(f"Expected {self.n_inputs} arrays, got {len(arrays)}.")
        import math

        exponent = 0.0
        if self.single is not None:
            out = self._einsum(self.single, arrays[0])
            return (out, exponent) if strip else out
        live = dict(enumerate(arrays))
        for i, eq in self.pre.items():
            live[i] = self._einsum(eq, live[i])
        for out_id, a, b, axes, eq, perm in self.schedule:
            x, y = live.pop(a), live.pop(b)
            if axes is not None:
                z = self._tensordot(x, y, axes)
                if perm:
                    z = self._transpose(z, perm)
            else:
                z = self._einsum(eq, x, y)
            if strip:
                # contract.py:816-829
                fac = float(abs(z).max())
                if check_zero and fac == 0.0:
                    return 0.0, float("-inf")
                exponent += math.log10(fac)
                z = z / fac
            live[out_id] = z
        out = live[self.root]
        return (out, exponent) if strip else out


def _chunk_index(tree, loc):
    """Position of a slice's output inside the full result tensor: sliced
    output indices are fixed (a projected one sits at 0 of its size-1 axis),
    the others span their axis."""
    index = []
    for ix in tree.output:
        si = tree.sliced_inds.get(ix)
        if si is None:
            index.append(slice(None))
        else:
            index.append(0 if si.project is not None else loc[ix])
    return tuple(index)


def _sliced_twin(tree):
    """The same contraction schedule seen from inside one slice: sliced
    indices are removed from every term, nothing is sliced."""
    from .tree import ContractionTree

    twin = ContractionTree(
        tree.get_inputs_sliced(), tree.get_output_sliced(), tree.size_dict
    )
    twin.children = dict(tree.children)
    twin._extent = dict(tree._extent)
    twin._next_ssa = tree._next_ssa
    # indices sliced away no longer count as appearances
    # (explicit index orders -- ``sort_contraction_indices`` -- are part of the schedule)
    for node, d in tree._info.items():
        if "inds" in d:
            twin._info.setdefault(node, {})["inds"] = d["inds"]
    return twin


def make_contractor(
    tree,
    order=None,
    prefer_einsum=False,
    strip_exponent=False,
    check_zero=False,
    implementation=None,
    autojit=False,
    progbar=False,
    handle_slicing=False,
):
    """Reference ``make_contractor`` (contract.py:925-1006) with the MI355X
    executor as the only engine.  ``prefer_einsum`` and ``autojit`` are
    accepted for signature compatibility and have no effect: the plan never
    distinguishes tensordot from einsum steps and is already compiled."""
    if isinstance(implementation, (tuple, list)):
        # per-op plug-in (contract.py:775-776): the caller's (einsum, tensordot)
        return PerOpContractor(
            tree if (handle_slicing or not tree.sliced_inds) else _sliced_twin(tree),
            implementation, order=order, prefer_einsum=prefer_einsum,
            strip_exponent=strip_exponent, check_zero=check_zero,
        )
    if implementation not in (None, "auto", "hip"):
        raise ValueError(
            f"implementation={implementation!r} is not available: cotengra_amd "
            "executes whole trees with its HIP backend ('hip') or with a caller's "
            "(einsum, tensordot) pair."
        )
    return HipContractor(
        tree,
        order=order,
        strip_exponent=strip_exponent,
        check_zero=check_zero,
        progbar=progbar,
        handle_slicing=handle_slicing,
    )


def _tree_contractor(tree, order=None):
    """Cached slicing-aware contractor of a tree."""
    key = ("hip-sliced", order if not callable(order) else id(order))
    try:
        return tree.contraction_cores[key]
    except KeyError:
        fn = tree.contraction_cores[key] = HipContractor(
            tree, order=order, handle_slicing=True
        )
        return fn


def contract_tree(
    tree,
    arrays,
    order=None,
    prefer_einsum=False,
    strip_exponent=False,
    check_zero=False,
    backend=None,
    implementation=None,
    autojit="auto",
    progbar=False,
):
    """``ContractionTree.contract`` (core.py:3943-4030): every slice runs on
    the device and accumulates into the resident output tensor."""
    if isinstance(implementation, (tuple, list)):
        # per-op plug-in: the host walks slices and steps (core.py:4002-4030)
        core = tree.get_contractor(
            order=order, prefer_einsum=prefer_einsum, strip_exponent=strip_exponent is not False,
            check_zero=check_zero, implementation=implementation,
        )
        if not tree.sliced_inds:
            return core(*arrays)
        slices = (core(*tree.slice_arrays(arrays, i)) for i in range(tree.multiplicity))
        return gather_slices(tree, slices, progbar=progbar)
    if implementation not in (None, "auto", "hip"):
        raise ValueError(f"implementation={implementation!r} is not available.")
    fn = _tree_contractor(tree, order)
    return fn(
        *arrays,
        strip_exponent=strip_exponent is not False,
        check_zero=check_zero,
        progbar=progbar,
    )


def contract_slice(tree, arrays, i, **kwargs):
    """``ContractionTree.contract_slice`` (core.py:3821-3823)."""
    order = kwargs.pop("order", None)
    strip_exponent = kwargs.pop("strip_exponent", False)
    check_zero = kwargs.pop("check_zero", False)
    implementation = kwargs.pop("implementation", None)
    prefer_einsum = kwargs.pop("prefer_einsum", False)
    for k in ("backend", "autojit", "progbar"):
        kwargs.pop(k, None)
    if kwargs:
        raise TypeError(f"Unknown keyword arguments: {kwargs}.")
    if not 0 <= i < tree.multiplicity:
        raise IndexError(f"slice {i} out of range [0, {tree.multiplicity})")
    if isinstance(implementation, (tuple, list)):
        return tree.contract_core(
            tree.slice_arrays(arrays, i), order=order, prefer_einsum=prefer_einsum,
            strip_exponent=strip_exponent, check_zero=check_zero, implementation=implementation,
        )
    fn = _tree_contractor(tree, order)
    return fn.contract_slice(
        arrays, i, strip_exponent=strip_exponent is not False, check_zero=check_zero
    )


def _add_maybe_stripped(x, y):
    """Exponent-aware sum of two slice outputs (core.py:125-172)."""
    xt, yt = isinstance(x, tuple), isinstance(y, tuple)
    if not (xt or yt):
        return x + y
    xm, xe = x if xt else (x, 0.0)
    ym, ye = y if yt else (y, 0.0)
    e = max(xe, ye)
    return xm * 10 ** (xe - e) + ym * 10 ** (ye - e), e


def gather_slices(tree, slices, backend=None, progbar=False):
    """Host-side gather of explicitly computed slice outputs
    (core.py:3825-3882); ``tree.contract`` never needs it because the device
    accumulates, but it is part of the public surface."""
    if progbar:
        prog = _Progress(progbar, tree.nslices)

        def counted(it):
            try:
                for x in it:
                    yield x
                    prog.update(1)
            finally:
                prog.close()

        slices = counted(slices)
    output_pos = {
        ix: i for i, ix in enumerate(tree.output) if ix in tree.sliced_inds
    }
    if not output_pos:
        import functools

        return functools.reduce(_add_maybe_stripped, slices)
    chunks = {}
    for i, s in enumerate(slices):
        ks = tree.slice_key(i)
        key = tuple(ks[ix] for ix in output_pos)
        chunks[key] = _add_maybe_stripped(chunks[key], s) if key in chunks else s
    if isinstance(next(iter(chunks.values())), tuple):
        emax = max(v[1] for v in chunks.values())
        chunks = {k: m * 10 ** (e - emax) for k, (m, e) in chunks.items()}
    else:
        emax = None
    first = next(iter(chunks.values()))
    if _is_torch(first):
        import torch

        stack_fn = lambda arrs, ax: torch.stack(arrs, ax)  # noqa: E731
    else:
        stack_fn = lambda arrs, ax: np.stack(arrs, ax)  # noqa: E731

    def stack(loc, remaining):
        if not remaining:
            return chunks[loc]
        arrs = [
            stack(loc + (d,), remaining[1:])
            for d in tree.sliced_inds[remaining[0]].sliced_range
        ]
        return stack_fn(arrs, output_pos[remaining[0]] - len(loc))

    result = stack((), tuple(output_pos))
    return (result, emax) if emax is not None else result


def gen_output_chunks(tree, arrays, with_key=False, progbar=False, **contract_opts):
    """core.py:3884-3941: one chunk per combination of outer sliced indices,
    inner slices summed on the device."""
    order = contract_opts.pop("order", None)
    fn = _tree_contractor(tree, order)
    stepsize = prod(si.size for si in tree.sliced_inds.values() if si.inner)
    st = fn.setup(*arrays)
    ex = st["exec"]
    ex.set_strip_exponent(False)
    prog = _Progress(progbar, tree.nslices)
    st["pinned"] = st.get("pinned", 0) + 1   # (a suspended generator keeps its executor)
    try:
        for o in range(tree.nslices // stepsize):
            with fn._lock:
                ex.zero_result()
                ex.run_slices(o * stepsize, stepsize, 1)
                loc = tree.slice_key(o * stepsize)
                index = _chunk_index(tree, loc)
                chunk = fn._finish(st, False, False, index=index)
            if prog.active:
                prog.update(stepsize)
            if with_key:
                yield chunk, {ix: x for ix, x in loc.items() if ix in tree.output}
            else:
                yield chunk
    finally:
        st["pinned"] -= 1
        prog.close()


def benchmark_tree(
    tree, dtype="float64", max_time=60, min_reps=3, max_reps=100, warmup=True,
    **contract_opts,
):
    """``ContractionTree.benchmark`` (core.py:4092-4164) on the device: the
    inputs are made resident once, then ``contract_slice`` work is timed."""
    from .utils import make_arrays_from_inputs

    arrays = make_arrays_from_inputs(tree.inputs, tree.size_dict, dtype=dtype)
    fn = _tree_contractor(tree, contract_opts.pop("order", None))
    st = fn.setup(*arrays)
    ex = st["exec"]
    ex.set_strip_exponent(False)
    # slices are timed the way ``contract`` runs them: as many per launch sequence as
    # the executor batches (1 for wide trees, up to 64 for narrow ones)
    nb = max(1, min(int(ex.batch), tree.nslices))
    nstart = max(tree.nslices - nb + 1, 1)
    for i in range(int(warmup)):
        ex.run_slices((i * nb) % nstart, nb, 1)
    ex.sync()
    t0 = ti = time.time()
    i = 0
    while (ti - t0 < max_time) or (i < min_reps):
        ex.run_slices((i * nb) % nstart, nb, 1)
        ex.sync()
        ti = time.time()
        i += 1
        if i >= max_reps:
            break
    time_per_slice = (ti - t0) / (i * nb)
    est_time_total = time_per_slice * tree.nslices
    return {
        "time_per_slice": time_per_slice,
        "est_time_total": est_time_total,
        "est_gigaflops": tree.total_flops(dtype=dtype) / (1e9 * est_time_total),
    }


# ---------------------------------------------------------------------- #
# checkpoint / resume of a sliced run
# ---------------------------------------------------------------------- #
#
# The reference sums slices one after another into a running result
# (``gather_slices``, core.py:3842-3844; ``contract_mpi``, core.py:4073-4076) and
# keeps nothing else between slices, so (how many of my slices are done, the
# partial sum[, its exponent]) is the complete state of a run.  A Sycamore m20
# amplitude is days of GPU time: the state is written every so often and a
# restarted process continues from it -- bit-identically, because the slices
# are added in the same order onto the same bits.


def tree_signature(tree, dtype, rank=0, world=1, strip_exponent=False, check_zero=False, order=None,
                   arrays=None, groups=None):
    """Digest of everything a partial sum depends on: network, schedule,
    slicing, element type, which share of the slices, stripping options -- and,
    with ``arrays``, the input tensors themselves (``inputs_digest``): the same
    Sycamore tree contracted for another bitstring must not resume this sum."""
    import hashlib
    import json

    doc = {
        "inputs": [list(map(str, t)) for t in tree.inputs],
        "output": list(map(str, tree.output)),
        "sizes": sorted((str(k), int(v)) for k, v in tree.size_dict.items()),
        "ssa_path": [list(p) for p in tree.get_ssa_path(order=order)] if tree.N > 1 else [],
        "sliced": [[str(si.ind), si.project] for si in tree.sliced_inds.values()],
        "dtype": str(dtype),
        "share": [int(rank), int(world)],
        "strip": [bool(strip_exponent), bool(check_zero)],
        "arrays": inputs_digest(arrays) if arrays is not None else None,
    }
    if groups:   # (the order in which a rank's slices are summed follows the slice groups)
        doc["groups"] = [str(ix) for ix in groups]
    return hashlib.sha256(json.dumps(doc, sort_keys=True).encode()).hexdigest()


def save_checkpoint(path, signature, done, result, exponent=0.0, zero=False):
    """Atomically replace ``path`` (write to a sibling, fsync, rename): a crash
    while writing leaves the previous checkpoint intact."""
    import os
    import tempfile

    d = os.path.dirname(os.path.abspath(path)) or "."
    fd, tmp = tempfile.mkstemp(prefix=os.path.basename(path) + ".", suffix=".tmp", dir=d)
    try:
        with os.fdopen(fd, "wb") as f:
            np.savez(
                f, signature=np.array(signature), done=np.int64(done), result=np.asarray(result),
                exponent=np.float64(exponent), zero=np.bool_(zero),
            )
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, path)
    except BaseException:
        try:
            os.unlink(tmp)
        except OSError:
            pass
        raise


def load_checkpoint(path, signature):
    """``(done, result, exponent, zero)`` or None when there is no checkpoint;
    ``ValueError`` when the file belongs to a different contraction."""
    import os

    if not os.path.exists(path):
        return None
    with np.load(path, allow_pickle=False) as z:
        if str(z["signature"]) != signature:
            raise ValueError(
                f"checkpoint {path} was written for a different tree / dtype / rank layout / "
                "set of input tensors (signature mismatch); remove it to start over."
            )
        return int(z["done"]), z["result"].copy(), float(z["exponent"]), bool(z["zero"])


def contract_resumable(
    tree, arrays, checkpoint, every=64, order=None, strip_exponent=False, check_zero=False,
    rank=0, world=1, stop_after=None, keep=False, progbar=False,
):
    """``tree.contract(arrays)`` that survives being killed.

    This rank's slices (``rank, rank + world, ...``: ``world=1`` is all of
    them) run in chunks of ``every``; after each chunk the partial sum is
    downloaded (``ctg_exec_get_state``) and written to ``checkpoint``.  If the
    file already exists the run restores it (``ctg_exec_set_state``) and
    continues with the first slice not yet summed.  ``stop_after=n`` ends this
    call after at most ``n`` more slices (returns None unless that completed
    the run) -- what a job-time limit or a test uses.  The file is removed when
    the run completes unless ``keep`` (that removal is housekeeping, not a
    guard: what ties a file to a run is its signature, which covers the tree,
    the options, the rank's share AND the bytes of the input tensors).  Returns
    what ``contract`` returns (this rank's share when ``world > 1``: feed it to
    the collective)."""
    import os

    fn = _tree_contractor(tree, order)
    with fn._lock:
        return _contract_resumable_locked(
            fn, tree, arrays, checkpoint, every, order, strip_exponent, check_zero, rank, world,
            stop_after, keep, progbar,
        )


def _contract_resumable_locked(fn, tree, arrays, checkpoint, every, order, strip_exponent, check_zero,
                               rank, world, stop_after, keep, progbar):
    import os

    st = fn.setup(*arrays)
    ex = st["exec"]
    ex.set_strip_exponent(strip_exponent, check_zero)
    plan = st["plan"]
    # this rank's share is what the library deals it (Plan.share_units: whole slice groups rank, rank + world,
    # ... -- single slices without group indices); the count of the checkpoint runs along that order and
    # chunks are whole units
    units, gs = plan.share_units(rank, world)
    total = units * gs
    grouped = gs > 1
    every = gs * max(1, -(-int(every) // gs))

    def ids_at(start, n):
        u0, u1 = start // gs, -(-(start + n) // gs)
        return plan.rank_slice_ids(rank, world, u0, u1 - u0)[start - u0 * gs: start - u0 * gs + n]
    sig = tree_signature(tree, st["plan"].dtype, rank, world, strip_exponent, check_zero, order,
                         arrays=arrays, groups=plan.group_inds if grouped else None)
    saved = load_checkpoint(checkpoint, sig)
    if saved is None:
        done = 0
        ex.zero_result()
    else:
        done, result, exponent, zero = saved
        if not 0 <= done <= total:
            raise ValueError(f"checkpoint {checkpoint} claims {done} of {total} slices")
        ex.set_state(result, exponent, zero)
    budget = total - done if stop_after is None else min(int(stop_after), total - done)
    prog = _Progress(progbar, total)
    if prog.active and done:
        prog.update(done)
    try:
        while budget > 0:
            n = min(int(every), budget)
            if done % gs == 0 and n % gs == 0:
                ex.run_share(rank, world, done // gs, n // gs)
            else:   # (a run stopped inside a group by ``stop_after``)
                ex.run_slice_list(ids_at(done, n))
            done += n
            budget -= n
            # (the running sum in the precision the executor carries it in: double for single-precision trees)
            result, exponent, zero = ex.get_state_wide()
            save_checkpoint(checkpoint, sig, done, result, exponent, zero)
            if prog.active:
                prog.update(n)
    finally:
        prog.close()
    if done < total:
        return None
    out = fn._finish(st, strip_exponent, check_zero)
    if not keep and os.path.exists(checkpoint):
        os.unlink(checkpoint)
    return out
