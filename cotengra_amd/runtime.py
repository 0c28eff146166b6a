"""ctypes binding of ``libctg_hip.so`` (C ABI: ``include/ctg_hip.h``).

The HIP library is the only execution engine: if it is missing this module
raises immediately -- there is no CPU fallback anywhere in the package.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

_LIB_PATH = os.environ.get("CTG_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libctg_hip.so")
ABI_VERSION = 7

# every symbol include/ctg_hip.h declares
SYMBOLS = (
    "ctg_abi_version",
    "ctg_last_error",
    "ctg_plan_create",
    "ctg_plan_destroy",
    "ctg_plan_nslices",
    "ctg_plan_workspace_bytes",
    "ctg_exec_create",
    "ctg_exec_destroy",
    "ctg_exec_set_stream",
    "ctg_exec_upload_inputs_host",
    "ctg_exec_upload_inputs_device",
    "ctg_exec_zero_result",
    "ctg_exec_set_strip_exponent",
    "ctg_exec_set_stem_arithmetic",
    "ctg_exec_get_exponent",
    "ctg_exec_run_slices",
    "ctg_exec_run_slice_list",
    "ctg_plan_share_units",
    "ctg_plan_share_slice_ids",
    "ctg_exec_run_share",
    "ctg_exec_slice_batch",
    "ctg_exec_device_bytes",
    "ctg_stem_triple_instantiated",
    "ctg_exec_launch_count",
    "ctg_exec_profile_slice",
    "ctg_exec_step_kernel",
    "ctg_exec_sync",
    "ctg_exec_result_ptr",
    "ctg_exec_download_result",
    "ctg_exec_download_arena",
    "ctg_exec_get_state",
    "ctg_exec_set_state",
    "ctg_exec_state_dtype",
    "ctg_exec_get_state_wide",
    "ctg_exec_set_state_wide",
    "ctg_comm_get_unique_id",
    "ctg_comm_init",
    "ctg_comm_info",
    "ctg_comm_destroy",
    "ctg_exec_reduce",
    "ctg_path_greedy",
    "ctg_slice_greedy",
    "ctg_subtree_reconfigure",
    "ctg_subtree_reconfigure_timed",
)


class CtgError(RuntimeError):
    """A call into libctg_hip.so failed."""


class CommError(CtgError):
    """RCCL is missing or a collective failed (CTG_E_COMM)."""


class PlanDesc(C.Structure):
    """Mirror of ``ctg_plan_desc``."""

    _fields_ = [
        ("dtype", C.c_int32),
        ("n_inputs", C.c_int64),
        ("input_sizes", C.POINTER(C.c_int64)),
        ("input_offsets", C.POINTER(C.c_int64)),
        ("inputs_elems", C.c_int64),
        ("arena_elems", C.c_int64),
        ("result_elems", C.c_int64),
        ("n_steps", C.c_int64),
        ("steps", C.POINTER(C.c_int64)),
        ("n_table_words", C.c_int64),
        ("tables", C.POINTER(C.c_int64)),
        ("n_sliced", C.c_int64),
        ("slice_sizes", C.POINTER(C.c_int64)),
        ("slice_fixed", C.POINTER(C.c_int64)),
        ("slice_strides", C.POINTER(C.c_int64)),
        ("slice_group", C.POINTER(C.c_int64)),
    ]


_lib = None


def lib_path():
    return _LIB_PATH


def load():
    """Load the shared library (once) and declare its prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). "
            "cotengra_amd has no CPU fallback."
        )
    # PyTorch bundles its own libamdhip64.  The host layer uses torch for device
    # tensors, streams and RCCL, so torch's runtime must be the one this library
    # binds to: loading ours first would bring a second HIP runtime into the
    # process ("no ROCm-capable device" on the later one).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(_LIB_PATH)
    i64p = C.POINTER(C.c_int64)
    vp = C.c_void_p
    lib.ctg_abi_version.restype = C.c_int
    lib.ctg_last_error.restype = C.c_char_p
    protos = {
        "ctg_plan_create": [C.POINTER(PlanDesc), C.POINTER(vp)],
        "ctg_plan_destroy": [vp],
        "ctg_plan_nslices": [vp, i64p],
        "ctg_plan_workspace_bytes": [vp, i64p],
        "ctg_exec_create": [vp, C.c_int, vp, vp, C.POINTER(vp)],
        "ctg_exec_destroy": [vp],
        "ctg_exec_set_stream": [vp, vp],
        "ctg_exec_upload_inputs_host": [vp, C.POINTER(vp)],
        "ctg_exec_upload_inputs_device": [vp, C.POINTER(vp)],
        "ctg_exec_zero_result": [vp],
        "ctg_exec_set_strip_exponent": [vp, C.c_int, C.c_int],
        "ctg_exec_set_stem_arithmetic": [vp, C.c_int],
        "ctg_exec_get_exponent": [vp, C.POINTER(C.c_double), C.POINTER(C.c_int)],
        "ctg_exec_run_slices": [vp, C.c_int64, C.c_int64, C.c_int64],
        "ctg_exec_run_slice_list": [vp, i64p, C.c_int64],
        "ctg_plan_share_units": [vp, C.c_int64, C.c_int64, i64p, i64p],
        "ctg_plan_share_slice_ids": [vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, i64p],
        "ctg_exec_run_share": [vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64],
        "ctg_exec_slice_batch": [vp, i64p],
        "ctg_exec_device_bytes": [vp, i64p],
        "ctg_stem_triple_instantiated": [C.c_int] * 9,
        "ctg_exec_launch_count": [vp, i64p, i64p],
        "ctg_exec_profile_slice": [vp, C.c_int64, C.POINTER(C.c_float)],
        "ctg_exec_step_kernel": [vp, C.c_int64, C.c_char_p, C.c_int64],
        "ctg_exec_sync": [vp],
        "ctg_exec_result_ptr": [vp, C.POINTER(vp)],
        "ctg_exec_download_result": [vp, vp],
        "ctg_exec_download_arena": [vp, C.c_int64, C.c_int64, vp],
        "ctg_exec_get_state": [vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_int)],
        "ctg_exec_set_state": [vp, vp, C.c_double, C.c_int],
        "ctg_exec_state_dtype": [vp, C.POINTER(C.c_int)],
        "ctg_exec_get_state_wide": [vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_int)],
        "ctg_exec_set_state_wide": [vp, vp, C.c_double, C.c_int],
        "ctg_comm_get_unique_id": [vp],
        "ctg_comm_init": [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)],
        "ctg_comm_info": [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
        "ctg_comm_destroy": [vp],
        "ctg_exec_reduce": [vp, vp, C.c_int],
        "ctg_path_greedy": [C.c_int64, i64p, i64p, C.c_int64, i64p, C.c_int64, C.POINTER(C.c_double),
                            C.c_double, C.c_double, C.c_int64, C.c_uint64, i64p],
        "ctg_slice_greedy": [C.c_int64, i64p, i64p, C.c_int64, i64p, C.c_int64, C.POINTER(C.c_double),
                             i64p, C.c_double, C.c_int, C.c_int64, i64p, i64p],
        "ctg_subtree_reconfigure": [C.c_int64, i64p, i64p, C.c_int64, i64p, C.c_int64, C.POINTER(C.c_double),
                                    i64p, C.c_int64, C.c_int64, C.c_double, i64p],
        "ctg_subtree_reconfigure_timed": [C.c_int64, i64p, i64p, C.c_int64, i64p, C.c_int64,
                                          C.POINTER(C.c_double), i64p, C.c_int64, C.c_int64,
                                          C.POINTER(C.c_double), C.c_int64, C.c_double, i64p],
    }
    for name, argtypes in protos.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    if lib.ctg_abi_version() != ABI_VERSION:
        raise ImportError(
            f"libctg_hip.so ABI {lib.ctg_abi_version()} != expected {ABI_VERSION}; rebuild."
        )
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        msg = load().ctg_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(msg)
        if rc == -3:
            raise MemoryError(msg)
        if rc == -5:
            raise CommError(msg)
        raise CtgError(f"[{rc}] {msg}")


def _i64p(arr):
    return arr.ctypes.data_as(C.POINTER(C.c_int64))


class DevicePlan:
    """A validated plan inside the native library (host memory only)."""

    def __init__(self, plan):
        self.plan = plan
        lib = load()
        s = plan.serialise()
        self._keep = s  # the C side deep-copies, but keep until created
        d = PlanDesc()
        d.dtype = s["dtype"]
        d.n_inputs = len(s["input_sizes"])
        d.input_sizes = _i64p(s["input_sizes"])
        d.input_offsets = _i64p(s["input_offsets"])
        d.inputs_elems = plan.inputs_elems
        d.arena_elems = s["arena_elems"]
        d.result_elems = s["result_elems"]
        d.n_steps = s["n_steps"]
        d.steps = _i64p(s["steps"])
        d.n_table_words = len(s["tables"])
        d.tables = _i64p(s["tables"])
        d.n_sliced = len(s["slice_sizes"])
        d.slice_sizes = _i64p(s["slice_sizes"])
        d.slice_fixed = _i64p(s["slice_fixed"])
        d.slice_strides = _i64p(s["slice_strides"])
        d.slice_group = _i64p(s["slice_group"])
        handle = C.c_void_p()
        _check(lib.ctg_plan_create(C.byref(d), C.byref(handle)))
        self.handle = handle
        self._keep = None

    @property
    def nslices(self):
        n = C.c_int64()
        _check(load().ctg_plan_nslices(self.handle, C.byref(n)))
        return n.value

    def share_units(self, rank=0, world=1):
        """``(units, slices per unit)`` of ``rank``'s share (``ctg_plan_share_units``): whole slice
        groups ``rank, rank + world, ...``; single slices for a plan without group indices."""
        u, g = C.c_int64(), C.c_int64()
        _check(load().ctg_plan_share_units(self.handle, int(rank), int(world), C.byref(u), C.byref(g)))
        return u.value, g.value

    def share_slice_ids(self, rank=0, world=1, unit_first=0, unit_count=-1):
        """Slice ids of the units ``[unit_first, unit_first + unit_count)`` of ``rank``'s share."""
        units, gsize = self.share_units(rank, world)
        n = units - unit_first if unit_count < 0 else unit_count
        out = np.empty(max(n, 0) * gsize, dtype=np.int64)
        _check(load().ctg_plan_share_slice_ids(self.handle, int(rank), int(world), int(unit_first), int(n), _i64p(out)))
        return out

    def workspace_bytes(self):
        out = (C.c_int64 * 4)()
        _check(load().ctg_plan_workspace_bytes(self.handle, out))
        return dict(zip(("inputs", "arena", "result", "tables"), out))

    def close(self):
        if getattr(self, "handle", None):
            load().ctg_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Executor:
    """One GPU's resident state for a plan: inputs, arena, tables, result."""

    def __init__(self, dplan, device=0, stream=0, result_ptr=None):
        self.dplan = dplan
        self.plan = dplan.plan
        self.device = int(device)
        handle = C.c_void_p()
        _check(
            load().ctg_exec_create(
                dplan.handle,
                self.device,
                C.c_void_p(int(stream) if stream else None),
                C.c_void_p(int(result_ptr) if result_ptr else None),
                C.byref(handle),
            )
        )
        self.handle = handle

    def set_stream(self, stream):
        """Enqueue all later work on ``stream`` (a raw ``hipStream_t`` value)."""
        _check(load().ctg_exec_set_stream(self.handle, C.c_void_p(int(stream) if stream else None)))

    # -- inputs ---------------------------------------------------------- #

    def upload_host(self, arrays):
        """``arrays``: numpy arrays; converted to the plan dtype, C order."""
        dt = np.dtype(self.plan.dtype)
        keep = [np.ascontiguousarray(x, dtype=dt) for x in arrays]
        self._check_sizes([x.size for x in keep])
        ptrs = (C.c_void_p * len(keep))(*[x.ctypes.data for x in keep])
        _check(load().ctg_exec_upload_inputs_host(self.handle, ptrs))

    def upload_device(self, ptrs, sizes):
        """``ptrs``: raw device addresses of contiguous tensors already in the
        plan dtype."""
        self._check_sizes(sizes)
        arr = (C.c_void_p * len(ptrs))(*[int(p) for p in ptrs])
        _check(load().ctg_exec_upload_inputs_device(self.handle, arr))

    def _check_sizes(self, sizes):
        want = list(self.plan.input_sizes)
        if list(sizes) != want:
            raise ValueError(
                f"Input sizes {list(sizes)} do not match the plan's {want}."
            )

    # -- execution --------------------------------------------------------- #

    def zero_result(self):
        _check(load().ctg_exec_zero_result(self.handle))

    def set_strip_exponent(self, strip, check_zero=False):
        _check(load().ctg_exec_set_strip_exponent(self.handle, int(bool(strip)), int(bool(check_zero))))

    STEM_ARITHMETICS = {"fp32": 0, "bf16x3": 1, "fp16x2": 2}

    def set_stem_arithmetic(self, mode):
        """Arithmetic of the fused stem pairs (include/ctg_hip.h): ``"fp32"`` / ``False`` / 0 -- fp32 products on
        the fp32 matrix cores; ``"bf16x3"`` / 1 -- operands split exactly into three bf16 limbs, six products;
        ``"fp16x2"`` / 2 -- two fp16 limbs under per-tensor power-of-two scales, three products (the
        executor's default since round 6).  ``True`` keeps its old meaning, ``"bf16x3"``."""
        if mode is True:
            mode = 1
        elif mode is False:
            mode = 0
        elif isinstance(mode, str):
            mode = self.STEM_ARITHMETICS[mode]
        _check(load().ctg_exec_set_stem_arithmetic(self.handle, int(mode)))

    def get_exponent(self):
        e, z = C.c_double(), C.c_int()
        _check(load().ctg_exec_get_exponent(self.handle, C.byref(e), C.byref(z)))
        return e.value, bool(z.value)

    def run_slice_list(self, ids):
        """Contract the slices ``ids`` (any order, any subset) and add them to the result; with slice
        groups in the plan, group by group (``ctg_exec_run_slice_list``)."""
        arr = np.ascontiguousarray(ids, dtype=np.int64)
        _check(load().ctg_exec_run_slice_list(self.handle, arr.ctypes.data_as(C.POINTER(C.c_int64)), arr.size))

    def run_share(self, rank=0, world=1, unit_first=0, unit_count=-1):
        """Contract units ``[unit_first, unit_first + unit_count)`` of ``rank``'s share of the slices
        (``ctg_exec_run_share``: whole slice groups ``rank, rank + world, ...``; the round-robin of
        ``contract_mpi``, core.py:4070, for a plan without groups) and add them to the result."""
        _check(load().ctg_exec_run_share(self.handle, int(rank), int(world), int(unit_first), int(unit_count)))

    def run_slices(self, first=0, count=None, stride=1):
        if count is None:
            count = (self.plan.nslices - first + stride - 1) // stride
        _check(load().ctg_exec_run_slices(self.handle, first, count, stride))

    @property
    def batch(self):
        """Slices that share one launch sequence in ``run_slices`` (1 for wide trees)."""
        n = C.c_int64()
        _check(load().ctg_exec_slice_batch(self.handle, C.byref(n)))
        return n.value

    def device_bytes(self):
        """Device memory the executor holds (arena x slice batch, inputs, tables, result,
        scratch): ``ctg_exec_device_bytes``."""
        n = C.c_int64()
        _check(load().ctg_exec_device_bytes(self.handle, C.byref(n)))
        return n.value

    def launch_count(self):
        """``(steps, launches)`` of one slice: independent small steps share launches."""
        ns, nl = C.c_int64(), C.c_int64()
        _check(load().ctg_exec_launch_count(self.handle, C.byref(ns), C.byref(nl)))
        return ns.value, nl.value

    def profile_slice(self, slice_id=0):
        ms = (C.c_float * max(len(self.plan.steps), 1))()
        _check(load().ctg_exec_profile_slice(self.handle, slice_id, ms))
        return np.asarray(ms[: len(self.plan.steps)], dtype=np.float64)

    def step_kernels(self):
        """Kernel name per plan step (as rocprof reports it, shortened)."""
        buf = C.create_string_buffer(160)
        names = []
        for s in range(len(self.plan.steps)):
            _check(load().ctg_exec_step_kernel(self.handle, s, buf, 160))
            names.append(buf.value.decode())
        return names

    def sync(self):
        _check(load().ctg_exec_sync(self.handle))

    def result_ptr(self):
        p = C.c_void_p()
        _check(load().ctg_exec_result_ptr(self.handle, C.byref(p)))
        return p.value

    def download_result(self):
        out = np.empty(self.plan.result_shape, dtype=np.dtype(self.plan.dtype))
        _check(load().ctg_exec_download_result(self.handle, C.c_void_p(out.ctypes.data)))
        return out

    # -- checkpoint state (include/ctg_hip.h: ctg_exec_get_state / set_state) -- #

    def get_state(self):
        """``(partial_result, exponent, zero)`` after synchronising: everything
        a sliced run has accumulated so far."""
        out = np.empty(self.plan.result_shape, dtype=np.dtype(self.plan.dtype))
        e, z = C.c_double(), C.c_int()
        _check(load().ctg_exec_get_state(self.handle, C.c_void_p(out.ctypes.data), C.byref(e), C.byref(z)))
        return out, e.value, bool(z.value)

    def set_state(self, result, exponent=0.0, zero=False):
        """Restore what ``get_state`` / ``get_state_wide`` returned (instead of ``zero_result``): an array in
        the executor's state dtype -- the double-precision running sum of a single-precision sliced tree --
        continues the sum with the bits an uninterrupted run would carry; one in the plan's own dtype restarts
        it from the rounded values."""
        arr = np.asarray(result)
        wide = self.state_dtype()
        if arr.dtype == wide and wide != np.dtype(self.plan.dtype):
            arr = np.ascontiguousarray(arr)
            fn = load().ctg_exec_set_state_wide
        else:
            arr = np.ascontiguousarray(arr, dtype=np.dtype(self.plan.dtype))
            fn = load().ctg_exec_set_state
        if arr.size != int(np.prod(self.plan.result_shape, dtype=np.int64)):
            raise ValueError(f"state has {arr.size} elements, the plan's result {self.plan.result_shape}")
        _check(fn(self.handle, C.c_void_p(arr.ctypes.data), float(exponent), int(bool(zero))))

    def state_dtype(self):
        """Element type of the running sum (``ctg_exec_state_dtype``): float64 / complex128 for a
        float32 / complex64 sliced tree, else the plan's dtype."""
        d = C.c_int()
        _check(load().ctg_exec_state_dtype(self.handle, C.byref(d)))
        return np.dtype(("float32", "float64", "complex64", "complex128")[d.value])

    def get_state_wide(self):
        """``(running sum in state_dtype, exponent, zero)``: what a checkpoint stores."""
        out = np.empty(self.plan.result_shape, dtype=self.state_dtype())
        e, z = C.c_double(), C.c_int()
        _check(load().ctg_exec_get_state_wide(self.handle, C.c_void_p(out.ctypes.data), C.byref(e), C.byref(z)))
        return out, e.value, bool(z.value)

    # -- multi-GPU ---------------------------------------------------------- #

    def reduce(self, comm, root=None):
        """Sum the ranks' result tensors in place with RCCL on this executor's
        stream (``root=None``: all-reduce).  Reference ``contract_mpi``'s
        ``Allreduce`` / ``Reduce`` (core.py:4081, 4089)."""
        _check(load().ctg_exec_reduce(self.handle, comm.handle, -1 if root is None else int(root)))

    def download_arena(self, offset, n):
        out = np.empty(n, dtype=np.dtype(self.plan.dtype))
        _check(
            load().ctg_exec_download_arena(
                self.handle, offset, n, C.c_void_p(out.ctypes.data)
            )
        )
        return out

    def close(self):
        if getattr(self, "handle", None):
            load().ctg_exec_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """An RCCL communicator behind the C ABI (``ctg_comm_*``): one per process
    and GPU.  ``Comm.unique_id()`` on one rank, the 128 bytes handed to the
    others by any channel, then ``Comm(id, rank, world, device)`` collectively."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(load().ctg_comm_get_unique_id(buf))
        return buf.raw

    def __init__(self, unique_id, rank, world, device=0):
        if len(unique_id) != 128:
            raise ValueError("unique id must be 128 bytes")
        handle = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _check(load().ctg_comm_init(buf, int(rank), int(world), int(device), C.byref(handle)))
        self.handle = handle
        self.rank, self.world, self.device = int(rank), int(world), int(device)

    @classmethod
    def from_torch_group(cls, group=None, device=None):
        """Create the communicator over the ranks of a ``torch.distributed``
        group, using the group only to hand the unique id around."""
        import torch
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if device is None:
            device = torch.cuda.current_device()
        return cls(box[0], rank, world, device)

    def close(self):
        if getattr(self, "handle", None):
            load().ctg_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
