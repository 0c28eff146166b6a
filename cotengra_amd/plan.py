"""Compile a (sliced) contraction tree into a static device plan.

cotengra executes a tree eagerly: per pairwise step it materialises
``transpose -> reshape -> matmul -> reshape -> transpose`` through a generic
array API and allocates every intermediate (reference
``cotengra/contract.py:364-411, 791-832``).  The MI355X executor instead
compiles the tree ONCE into a flat list of *gather-GEMM* steps:

    C[rowC(R) + colC(n)] = alpha * sum_k A[rowA(R) + kA(k)] * B[rowB(R) + kB(k) + colB(n)]

where every index of the pairwise einsum belongs to exactly one of four
groups -- batch/kept-left (rows ``R``), contracted (``k``), kept-right
(``n``) -- and each operand's address is a *sum of per-group offset tables*.
No operand is ever permuted or reshaped in memory: the axis permutation the
reference performs with ``transpose`` + ``reshape`` copies is folded into the
offset tables and realised inside the GEMM kernel's (coalesced, LDS-staged)
load path.  The same classification as the reference's
``_parse_eq_to_batch_matmul`` (contract.py:168-329) is used -- batch =
on A, B and out; contracted = on A and B only; kept = on one operand and out
-- but its output is addressing metadata instead of array ops.

Slicing an index is likewise free: a sliced leaf is a strided view of the
full input resident in HBM whose base offset depends on the slice id
(reference ``core.py:3802-3819`` does the same with numpy views); the plan
records per-leaf strides of the sliced indices and the executor's prologue
kernel turns a slice id into base offsets on the device.

Intermediates live in one arena whose offsets are assigned here from the
traversal's liveness (the reference frees operands by ``temps.pop``,
contract.py:806-807; its peak model is core.py:1299-1316).

Steps come in three sharing classes: per slice; slice-invariant (below which no
sliced index occurs: once per upload); and, round 4, shared by a *slice group*
(``choose_slice_group``: below which none of a few chosen "group indices"
occurs -- once per group of slices that differ only in those, what per-slice
steps read of them kept outside the recycled part of the arena).  Consecutive
steps of a contraction stem are emitted as one fused step (``stem.py``).

The plan is serialised to flat int64 arrays (see ``Plan.serialise``) and
handed through the C ABI in ``include/ctg_hip.h``.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import math
import os

import numpy as np

from .utils import prod

# ---- constants shared with csrc/ctg_common.h ------------------------------ #

DTYPE_CODES = {"float32": 0, "float64": 1, "complex64": 2, "complex128": 3}
DTYPE_ITEMSIZE = {"float32": 4, "float64": 8, "complex64": 8, "complex128": 16}

KIND_SINGLE = 0  # out[o] = sum_s in[offO(o) + offS(s)]
KIND_PAIR = 1  # gather-GEMM, see module docstring
KIND_ACCUM = 2  # result[chunk + offR(o)] += src[o]
KIND_STEM2 = 3  # two consecutive pair steps of a stem as one launch (stem.py)

KERNEL_VALU = 0  # one thread per output element, any dtype / any shape
KERNEL_MFMA = 1  # matrix-core kernels (complex64 on fp32 MFMA, complex128 on fp64 MFMA)

SPACE_INPUTS = 0
SPACE_ARENA = 1
SPACE_RESULT = 2

STEP_WORDS = 48  # int64 words per serialised step record
W_STEM = 43  # STEM2 steps: word offset of the descriptor in the table blob (stem.serialise_stem)
W_LDS_COMP = 44  # member of an LDS-resident subtree (ldsrun.py): component id + 1
W_LDS_DESC = 45  # ... word offset of the component's descriptor in the table blob
LO_MAX = 4096  # target size of the fast ('lo') level of a row table
ARENA_ALIGN = 64  # elements; keeps every intermediate 256-B aligned
# trees whose largest intermediate is at most this are emitted level by level (compile_tree)
LEVEL_ORDER_MAX_ELEMS = 1 << 22
MAX_TENSOR_ELEMS = 1 << 36  # one complex128 tensor of this size is 1 TiB: beyond any single device

# MFMA kernel limits (see csrc/ctg_pair_mfma.hip)
MFMA_MAX_BATCH = 65535
# both operands at least this big (elements) => neither is cache resident
INTERLEAVE_MIN_ELEMS = 1 << 22


def _row_major_strides(shape):
    strides = [1] * len(shape)
    for i in range(len(shape) - 2, -1, -1):
        strides[i] = strides[i + 1] * shape[i + 1]
    return tuple(strides)


def group_table(extents, strides):
    """Flat offset table of an index group: entry ``i`` (row-major over
    ``extents``, last fastest) is ``sum_j digit_j(i) * strides[j]``."""
    table = np.zeros(1, dtype=np.int64)
    for d, s in zip(extents, strides):
        table = (table[:, None] + (np.arange(d, dtype=np.int64) * s)[None, :])
        table = table.reshape(-1)
    return table


def split_point(extents, lo_max=LO_MAX):
    """Number of trailing dims forming the 'lo' level of a two-level table:
    the longest suffix whose product stays <= ``lo_max`` (at least one dim if
    there is any)."""
    lo = 1
    n = 0
    for d in reversed(extents):
        if lo * d > lo_max and n > 0:
            break
        lo *= d
        n += 1
    return n, lo


@dataclass
class TensorRef:
    """Where a tensor lives and how it is laid out.

    ``inds`` may contain repeated labels only for raw leaves that still need
    a SINGLE (diag/trace) preprocessing step."""

    space: int
    offset: int  # static element offset inside the space
    leaf: int  # input number whose slice offset applies, or -1
    inds: tuple
    strides: tuple  # element strides per axis
    size: int  # number of addressable elements (for bounds checks)

    def stride_of(self, ix):
        """Total stride of label ``ix`` (sum over repeats; 0 if absent)."""
        return sum(s for i, s in zip(self.inds, self.strides) if i == ix)


@dataclass
class Step:
    kind: int
    kernel: int = KERNEL_VALU
    a: TensorRef = None
    b: TensorRef = None
    c: TensorRef = None
    # group extents
    R: int = 1  # rows = batch * M   (VALU) ; M (MFMA)
    Bt: int = 1  # batch (MFMA only; VALU folds it into R)
    K: int = 1
    N: int = 1
    # two-level row tables: lo_size, per-operand (hi, lo) arrays
    row_lo: int = 1
    rows: dict = field(default_factory=dict)  # 'A','B','C' -> (hi, lo)
    k_lo: int = 1
    k_tabs: dict = field(default_factory=dict)  # 'A','B' -> (hi, lo)
    n_tabs: dict = field(default_factory=dict)  # 'B','C' -> flat
    b_tabs: dict = field(default_factory=dict)  # 'A','B','C' -> flat (MFMA)
    # bookkeeping for rooflines / debugging
    macs: int = 0
    elems_rw: int = 0
    node: int = -1
    label: str = ""
    # index of the pair step that produced operand a / b (-1: input or
    # preprocessing output, whose scale factor is 1) -- strip_exponent
    a_prod: int = -1
    b_prod: int = -1
    # True if the step does not depend on any sliced input: it is executed once
    # per upload instead of once per slice and its output is never recycled
    invariant: bool = False
    # (round 4) does not depend on the GROUP indices of the plan (Plan.group_inds): computed once per group of
    # slices that differ only in those, its result kept where per-slice steps do not recycle it
    group: bool = False
    # STEM2 (stem.py): the second step's small operand, its producer, the tile
    # geometry + tables; elems_rw counts BOTH steps as if unfused (the roofline's
    # algorithmic bytes), elems_moved what the fused launch really moves
    b2: TensorRef = None
    b2_prod: int = -1
    stem: dict = None
    elems_moved: int = 0
    # (round 6) member of an LDS-resident subtree (ldsrun.py): index into Plan.lds_runs, -1 = none
    lds_comp: int = -1


class Arena:
    """First-fit free-list allocator over element offsets."""

    def __init__(self, align=ARENA_ALIGN):
        self.align = align
        self.free = []  # sorted list of (offset, size)
        self.top = 0
        self.peak = 0

    def _round(self, n):
        return (n + self.align - 1) // self.align * self.align

    def alloc(self, n):
        n = self._round(max(n, 1))
        for i, (off, sz) in enumerate(self.free):
            if sz >= n:
                if sz == n:
                    self.free.pop(i)
                else:
                    self.free[i] = (off + n, sz - n)
                return off
        # grow; merge with a trailing free block if it touches the top
        if self.free and self.free[-1][0] + self.free[-1][1] == self.top:
            off, sz = self.free.pop()
            self.top = off + n
        else:
            off = self.top
            self.top += n
        self.peak = max(self.peak, self.top)
        return off

    def release(self, off, n):
        n = self._round(max(n, 1))
        self.free.append((off, n))
        self.free.sort()
        merged = []
        for o, s in self.free:
            if merged and merged[-1][0] + merged[-1][1] == o:
                merged[-1] = (merged[-1][0], merged[-1][1] + s)
            else:
                merged.append((o, s))
        self.free = merged


class Plan:
    """A compiled contraction: steps + tables + slice metadata."""

    def __init__(self, dtype):
        if dtype not in DTYPE_CODES:
            raise ValueError(f"unsupported dtype {dtype}")
        self.dtype = dtype
        self.steps = []
        self.input_sizes = []  # elements per (unsliced) input
        self.input_offsets = []  # element offset of each input in the inputs space
        self.arena_elems = 0
        self.result_elems = 1
        self.result_shape = ()
        # slicing
        self.slice_sizes = []  # extent per sliced index (1 if projected)
        self.slice_fixed = []  # projected value or -1
        self.slice_strides = None  # (n_inputs + 1, n_sliced) element strides
        # slice groups (round 4): slices that differ only in these sliced indices share every step that does
        # not depend on them (Step.group); one flag per sliced index, in the order of slice_sizes
        self.group_inds = ()
        self.slice_group = []
        self.nslices = 1
        # LDS-resident subtrees (round 6, ldsrun.py): per component its shadow steps, LDS need and members
        self.lds_runs = []
        # accounting
        self.macs_per_slice = 0
        self.elems_rw_per_slice = 0
        # elements a slice really moves: less than elems_rw_per_slice when stem pairs
        # are fused (their intermediate stays on the chip)
        self.elems_moved_per_slice = 0

    # ------------------------------------------------------------------ #

    @property
    def itemsize(self):
        return DTYPE_ITEMSIZE[self.dtype]

    # ---- slice groups: which slices share the steps marked ``group`` ---------------------------------
    def _slice_digits(self):
        """``(stride in the slice id, size, is a group index)`` of every sliced index that is not
        projected, least significant first (the last sliced index varies fastest: tree.get_slice_strides)."""
        flags = list(self.slice_group) if len(self.slice_group) == len(self.slice_sizes) else [0] * len(self.slice_sizes)
        out, stride = [], 1
        for j in range(len(self.slice_sizes) - 1, -1, -1):
            if self.slice_fixed[j] >= 0:
                continue
            out.append((stride, int(self.slice_sizes[j]), bool(flags[j])))
            stride *= int(self.slice_sizes[j])
        return out

    @property
    def group_size(self):
        """Slices per group (1: the plan shares nothing between slices but the invariant steps)."""
        return prod(size for _, size, g in self._slice_digits() if g)

    def group_ids(self, g):
        """Slice ids of the ``g``-th group, ascending (``g`` in ``range(nslices // group_size)``: the digits
        of ``g`` are the values of the sliced indices that are not group indices)."""
        digits = self._slice_digits()
        base, rem = 0, int(g)
        for stride, size, is_group in digits:
            if not is_group:
                base += (rem % size) * stride
                rem //= size
        ids = [base]
        for stride, size, is_group in digits:
            if is_group:
                ids = [i + d * stride for i in ids for d in range(size)]
        return sorted(ids)

    def group_of(self, slice_id):
        """The group number (argument of ``group_ids``) ``slice_id`` belongs to."""
        # (a Python int of any size -- trees narrowed for tests have more than 2^63 slices -- or an
        # int64 array of slice ids)
        scalar = isinstance(slice_id, (int, np.integer))
        rem = int(slice_id) if scalar else np.asarray(slice_id, dtype=np.int64)
        g, mult = (0 if scalar else np.zeros_like(rem)), 1
        for stride, size, is_group in self._slice_digits():
            d = rem % size
            rem = rem // size
            if not is_group:
                g = g + d * mult
                mult *= size
        return g

    # ---- a rank's share of the slices (the library's rule: ctg_plan_share_units / ctg_exec_run_share) -----
    def share_units(self, rank=0, world=1):
        """``(units, slices per unit)`` of ``rank``'s share: the units ``rank, rank + world, ...`` where a
        unit is a whole slice group -- what a group shares is then computed once per group, on one rank --
        and a single slice for a plan without group indices, which is ``contract_mpi``'s round-robin
        (core.py:4068-4076)."""
        if world < 1 or not 0 <= rank < world:
            raise ValueError(f"rank {rank} of {world}")
        gs = int(self.group_size)
        n_units = int(self.nslices) // gs
        return (len(range(rank, n_units, world)), gs)

    def rank_slice_ids(self, rank=0, world=1, unit_first=0, unit_count=None):
        """Slice ids of (units ``[unit_first, unit_first + unit_count)`` of) ``rank``'s share, unit after
        unit, ascending inside a unit: an int64 array.  The shares of the ranks are disjoint, cover
        ``range(nslices)`` and differ by at most one unit."""
        units, gs = self.share_units(rank, world)
        if unit_count is None:
            unit_count = units - unit_first
        if unit_first < 0 or unit_count < 0 or unit_first + unit_count > units:
            raise ValueError(f"units [{unit_first}, +{unit_count}) outside the {units} of rank {rank}")
        g = rank + (unit_first + np.arange(unit_count, dtype=np.int64)) * world
        if gs == 1 or len(g) == 0:
            return g
        digits = self._slice_digits()
        base, rem = np.zeros_like(g), g.copy()
        for stride, size, is_group in digits:
            if not is_group:
                base += (rem % size) * stride
                rem //= size
        ids = base[:, None]
        for stride, size, is_group in digits:
            if is_group:
                ids = (ids[:, :, None] + (np.arange(size, dtype=np.int64) * stride)[None, None, :]).reshape(len(g), -1)
        return np.sort(ids, axis=1).reshape(-1)

    @property
    def macs_shared_per_group(self):
        """Multiply-adds of the steps a group computes once."""
        return sum(s.macs for s in self.steps if s.group)

    @property
    def is_complex(self):
        return self.dtype.startswith("complex")

    def flops_per_slice(self):
        """Real floating point operations per slice: 8 per complex
        multiply-add, 2 per real one."""
        return (8 if self.is_complex else 2) * self.macs_per_slice

    def bytes_per_slice(self):
        """Algorithmic bytes per slice: every operand read once and every
        result written once, permutes assumed fused."""
        return self.itemsize * self.elems_rw_per_slice

    # ------------------------------------------------------------------ #

    def serialise(self):
        """Flatten to ``(header, steps, tables)`` int64 arrays for the C ABI.

        tables is one blob; each step record stores word offsets into it.
        Layout of a step record (STEP_WORDS int64 words):

          0 kind            1 kernel
          2 a.space  3 a.offset  4 a.leaf
          5 b.space  6 b.offset  7 b.leaf
          8 c.space  9 c.offset 10 c.leaf
         11 R   12 Bt   13 K   14 N
         15 row_lo   16 row_hi_len
         17 rowA_hi 18 rowA_lo 19 rowB_hi 20 rowB_lo 21 rowC_hi 22 rowC_lo
         23 kA   24 kB   25 nB   26 nC
         27 bA   28 bB   29 bC
         30 a.size 31 b.size 32 c.size      (bounds, elements)
         33 macs    34 elems_rw   35 node
         36 k_lo  37 kA_hi  38 kB_hi  39 k_hi_len   (23/24 hold the lo level)
         40 a.producer step  41 b.producer step   (-1: scale factor 1)
         42 sharing class: 1 slice-invariant (run once per upload, output persistent), 2 shared by the
            slices of a group (Plan.group_inds: run once per group, what per-slice steps read of it kept)
         43 STEM2: word offset of the stem descriptor
         44 member of an LDS-resident subtree: component id + 1 (0: none)   45 word offset of the
            component's descriptor in the table blob (ldsrun.serialise_run)
         46.. reserved (0)
        """
        blobs = []
        cursor = 0

        def put(arr):
            nonlocal cursor
            arr = np.ascontiguousarray(arr, dtype=np.int64)
            off = cursor
            blobs.append(arr)
            cursor += arr.size
            return off

        zero = put(np.zeros(1, dtype=np.int64))  # shared all-zero table

        run_desc = []
        if self.lds_runs:
            from .ldsrun import serialise_run

            run_desc = [serialise_run(run, put) for run in self.lds_runs]

        recs = np.zeros((len(self.steps), STEP_WORDS), dtype=np.int64)
        for i, s in enumerate(self.steps):
            r = recs[i]
            r[0], r[1] = s.kind, s.kernel
            if s.kind == KIND_STEM2:
                from .stem import serialise_stem

                for base, t in ((2, s.a), (5, s.b), (8, s.c)):
                    r[base : base + 3] = (t.space, t.offset, t.leaf)
                r[11], r[12], r[13], r[14] = s.R, 1, s.K, s.N
                r[15] = r[16] = r[36] = r[39] = 1
                r[17:30] = -1
                r[30], r[31], r[32] = s.a.size, s.b.size, s.c.size
                r[33], r[34], r[35] = s.macs, s.elems_rw, s.node
                r[37] = r[38] = -1
                r[40], r[41] = s.a_prod, s.b_prod
                r[42] = 1 if s.invariant else (2 if s.group else 0)
                r[W_STEM] = serialise_stem(s, put)
                continue
            for base, t in ((2, s.a), (5, s.b), (8, s.c)):
                if t is not None:
                    r[base : base + 3] = (t.space, t.offset, t.leaf)
                else:
                    r[base : base + 3] = (-1, 0, -1)
            r[11], r[12], r[13], r[14] = s.R, s.Bt, s.K, s.N
            r[15] = s.row_lo
            hi_len = 1
            for j, key in enumerate("ABC"):
                if key in s.rows:
                    hi, lo = s.rows[key]
                    hi_len = len(hi)
                    r[17 + 2 * j] = put(hi)
                    r[18 + 2 * j] = put(lo)
                else:
                    r[17 + 2 * j] = -1
                    r[18 + 2 * j] = -1
            r[16] = hi_len
            r[36], r[39] = s.k_lo, 1
            r[23] = r[24] = r[37] = r[38] = zero
            for key, w_lo, w_hi in (("A", 23, 37), ("B", 24, 38)):
                if key in s.k_tabs:
                    hi, lo = s.k_tabs[key]
                    r[w_hi], r[w_lo] = put(hi), put(lo)
                    r[39] = len(hi)
            r[25] = put(s.n_tabs["B"]) if "B" in s.n_tabs else zero
            r[26] = put(s.n_tabs["C"]) if "C" in s.n_tabs else zero
            for j, key in enumerate("ABC"):
                r[27 + j] = put(s.b_tabs[key]) if key in s.b_tabs else zero
            r[30] = s.a.size if s.a is not None else 0
            r[31] = s.b.size if s.b is not None else 0
            r[32] = s.c.size if s.c is not None else 0
            r[33], r[34], r[35] = s.macs, s.elems_rw, s.node
            r[40], r[41] = s.a_prod, s.b_prod
            r[42] = 1 if s.invariant else (2 if s.group else 0)
            if s.lds_comp >= 0:
                r[W_LDS_COMP], r[W_LDS_DESC] = s.lds_comp + 1, run_desc[s.lds_comp]

        tables = np.concatenate(blobs) if blobs else np.zeros(1, np.int64)
        n_in = len(self.input_sizes)
        n_sl = len(self.slice_sizes)
        slice_strides = (
            np.ascontiguousarray(self.slice_strides, dtype=np.int64)
            if n_sl
            else np.zeros((n_in + 1, 0), dtype=np.int64)
        )
        return {
            "dtype": DTYPE_CODES[self.dtype],
            "input_sizes": np.asarray(self.input_sizes, dtype=np.int64),
            "input_offsets": np.asarray(self.input_offsets, dtype=np.int64),
            "arena_elems": int(self.arena_elems),
            "result_elems": int(self.result_elems),
            "steps": recs.reshape(-1),
            "n_steps": len(self.steps),
            "tables": tables,
            "slice_sizes": np.asarray(self.slice_sizes, dtype=np.int64),
            "slice_fixed": np.asarray(self.slice_fixed, dtype=np.int64),
            "slice_strides": slice_strides.reshape(-1),
            "slice_group": np.asarray(self.slice_group if len(self.slice_group) == n_sl else [0] * n_sl, dtype=np.int64),
        }

    def describe_steps(self):
        """Per-step rows for roofline tables (cf. the reference's
        ``print_contractions``, core.py:3508)."""
        rows = []
        for i, s in enumerate(self.steps):
            rows.append(
                {
                    "step": i,
                    "kind": ("single", "pair", "accum", "stem2")[s.kind],
                    "kernel": ("valu", "mfma")[s.kernel],
                    "R": s.R,
                    "Bt": s.Bt,
                    "K": s.K,
                    "N": s.N,
                    "macs": s.macs,
                    # algorithmic bytes (every operand read once, every result
                    # written once, per reference step -- a fused pair counts both of its
                    # steps) and the bytes the plan really moves (a fused pair: its big
                    # operand in, its result out, the two small operands)
                    "bytes": s.elems_rw * self.itemsize,
                    "bytes_moved": (s.elems_moved if s.kind == KIND_STEM2 else s.elems_rw) * self.itemsize,
                    "label": s.label,
                }
            )
        return rows


# --------------------------------------------------------------------------- #
# step builders
# --------------------------------------------------------------------------- #


def _rows_two_level(extents, stride_lists, lo_max=LO_MAX):
    """Two-level tables for a row group shared by several operands.

    Returns ``(lo_size, [(hi, lo), ...])`` with
    ``off(i) = hi[i // lo_size] + lo[i % lo_size]``."""
    nlo, lo_size = split_point(extents, lo_max)
    cut = len(extents) - nlo
    out = []
    for strides in stride_lists:
        hi = group_table(extents[:cut], strides[:cut])
        lo = group_table(extents[cut:], strides[cut:])
        out.append((hi, lo))
    return lo_size, out


def choose_kernel(dtype, Bt, M, K, N):
    """MFMA for complex64/float32 steps big enough to fill tiles; the VALU
    kernel for everything else (tiny leaves, outer products, Hadamards,
    skinny memory-bound steps, and the float64/complex128 parity mode)."""
    if Bt > MFMA_MAX_BATCH:
        return KERNEL_VALU
    if dtype != "complex64":
        # complex128 / float64 / float32 (csrc/ctg_pair_mfma_f64.hip): tiled
        # 16x16x4-MFMA kernels without split-K, so the output must be able to
        # fill the chip on its own
        if K >= 4 and N >= 8 and M >= 64 and M * N >= (1 << 16):
            return KERNEL_MFMA
        return KERNEL_VALU
    if K >= 4 and N >= 8 and M >= 32 and (M * N * K) >= (1 << 15):
        return KERNEL_MFMA
    # small result reduced over a very long contraction: the k-streaming kernel
    # (csrc: kstream_ok) splits K over the chip.  (Plain dot products stay on the
    # wavefront-reduction kernel: a 32 x 16 MFMA tile would be 1/32 full.)
    if Bt == 1 and K >= (1 << 16) and M <= 32 and N <= 32 and M * N >= 256:
        return KERNEL_MFMA
    # tall-skinny streaming kernel: HBM-bound, so padding N up to an MFMA tile
    # costs nothing (csrc/ctg_common.h: mfma_use_stream)
    if Bt == 1 and 2 <= K <= 128 and N <= 64 and M >= 8192:  # (stream or tiled, runtime picks)
        return KERNEL_MFMA
    # the same with a batch index: row-wise FMA kernel (csrc: pair_rowwise_kernel)
    if 2 <= K <= 32 and N <= 32 and M >= 8192:
        return KERNEL_MFMA
    return KERNEL_VALU


def pick_rows_operand(size_dict, l, r, out_inds):
    """``(A, B)``: the operand with more kept elements supplies the GEMM rows
    (ties: the left one) -- the rule of build_pair_step."""
    l_set, r_set, o_set = set(l.inds), set(r.inds), set(out_inds)
    rows_l = prod(size_dict[ix] for ix in dict.fromkeys(l.inds) if ix in o_set and ix not in r_set)
    rows_r = prod(size_dict[ix] for ix in dict.fromkeys(r.inds) if ix in o_set and ix not in l_set)
    return (r, l) if rows_r > rows_l else (l, r)


def build_pair_step(
    dtype, size_dict, l, r, out_inds, out_ref_factory, node=-1, force_kernel=None
):
    """Lower one pairwise einsum ``l, r -> out_inds`` to a PAIR step.

    ``out_ref_factory(inds, natural) -> TensorRef`` allocates the result;
    ``natural`` is the kernel-preferred index order ``[batch, M, N]`` and the
    factory may ignore it (the root must honour the user's output order).
    """
    l_set, r_set, o_set = set(l.inds), set(r.inds), set(out_inds)
    if not o_set <= (l_set | r_set):
        missing = o_set - (l_set | r_set)
        raise ValueError(f"Output indices {missing} not found on any input.")

    keep_l = [ix for ix in dict.fromkeys(l.inds) if ix in o_set and ix not in r_set]
    keep_r = [ix for ix in dict.fromkeys(r.inds) if ix in o_set and ix not in l_set]
    rows_l = prod(size_dict[ix] for ix in keep_l)
    rows_r = prod(size_dict[ix] for ix in keep_r)
    # the operand with more kept elements supplies the GEMM rows
    if rows_r > rows_l:
        A, B = r, l
        keep_a, keep_b = keep_r, keep_l
    else:
        A, B = l, r
        keep_a, keep_b = keep_l, keep_r
    a_set, b_set = set(A.inds), set(B.inds)

    a_order = list(dict.fromkeys(A.inds))
    b_order = list(dict.fromkeys(B.inds))
    batch = [ix for ix in a_order if ix in b_set and ix in o_set]
    # contracted / summed: everything not in the output (an index living on
    # one operand only is simply summed -- zero stride on the other operand)
    con = [ix for ix in a_order if ix not in o_set]
    con += [ix for ix in b_order if ix not in o_set and ix not in a_set]
    # Order of the contracted group = which k's share a k-step of the kernels.
    # Following A's memory order makes A's tile gathers contiguous; when B is
    # too large to live in cache as well, interleave the fastest-varying
    # contracted indices of both operands so that each k-step covers the low
    # address bits of BOTH (64-128 B runs on each side instead of 8 B on one).
    size_a = prod(size_dict[ix] for ix in a_order)
    size_b = prod(size_dict[ix] for ix in b_order)
    if min(size_a, size_b) >= INTERLEAVE_MIN_ELEMS and len(con) > 2:
        fast_a = sorted((ix for ix in con if ix in a_set), key=A.stride_of)
        fast_b = sorted((ix for ix in con if ix in b_set), key=B.stride_of)
        merged = []
        ia = ib = 0
        take_a = True
        while ia < len(fast_a) or ib < len(fast_b):
            src, pos = (fast_a, ia) if (take_a and ia < len(fast_a)) or ib >= len(fast_b) else (fast_b, ib)
            ix = src[pos]
            if src is fast_a:
                ia += 1
            else:
                ib += 1
            if ix not in merged:
                merged.append(ix)
                take_a = not take_a
        con = merged[::-1]  # tables put the LAST index fastest

    ext = lambda g: [size_dict[ix] for ix in g]  # noqa: E731
    Bt, M, K, N = (prod(ext(g)) for g in (batch, keep_a, con, keep_b))

    natural = tuple(batch) + tuple(keep_a) + tuple(keep_b)
    C = out_ref_factory(tuple(out_inds), natural)

    kernel = force_kernel
    if kernel is None:
        kernel = choose_kernel(dtype, Bt, M, K, N)

    step = Step(kind=KIND_PAIR, kernel=kernel, a=A, b=B, c=C, node=node)
    step.K, step.N = K, N
    step.k_lo, (step.k_tabs["A"], step.k_tabs["B"]) = _rows_two_level(
        ext(con),
        [[A.stride_of(ix) for ix in con], [B.stride_of(ix) for ix in con]],
    )
    step.n_tabs["B"] = group_table(ext(keep_b), [B.stride_of(ix) for ix in keep_b])
    step.n_tabs["C"] = group_table(ext(keep_b), [C.stride_of(ix) for ix in keep_b])

    if kernel == KERNEL_MFMA:
        step.R, step.Bt = M, Bt
        rows_g = keep_a
        lo, tabs = _rows_two_level(
            ext(rows_g),
            [
                [A.stride_of(ix) for ix in rows_g],
                [C.stride_of(ix) for ix in rows_g],
            ],
        )
        step.row_lo = lo
        step.rows["A"], step.rows["C"] = tabs
        for key, t in (("A", A), ("B", B), ("C", C)):
            step.b_tabs[key] = group_table(
                ext(batch), [t.stride_of(ix) for ix in batch]
            )
    else:
        step.R, step.Bt = Bt * M, 1
        rows_g = batch + keep_a
        lo, tabs = _rows_two_level(
            ext(rows_g),
            [
                [A.stride_of(ix) for ix in rows_g],
                [B.stride_of(ix) if ix in batch else 0 for ix in rows_g],
                [C.stride_of(ix) for ix in rows_g],
            ],
        )
        step.row_lo = lo
        step.rows["A"], step.rows["B"], step.rows["C"] = tabs

    step.macs = Bt * M * K * N
    step.elems_rw = (
        prod(ext(dict.fromkeys(A.inds)))
        + prod(ext(dict.fromkeys(B.inds)))
        + Bt * M * N
    )
    step.label = (
        f"{''.join(map(str, A.inds))},{''.join(map(str, B.inds))}"
        f"->{''.join(map(str, C.inds))}"
        if max(len(A.inds), len(B.inds)) <= 12
        else f"b{Bt} m{M} k{K} n{N}"
    )
    return step


def build_single_step(size_dict, src, out_inds, out_ref_factory, node=-1):
    """Lower a single-term einsum (diagonals, traces, sums, transposes;
    reference contract.py:62-119, 332-361) to a SINGLE step:
    ``out[o] = sum_s src[offO(o) + offS(s)]`` where repeated labels simply
    add their strides (a diagonal) and labels absent from the output are
    summed."""
    src_unique = list(dict.fromkeys(src.inds))
    o_set = set(out_inds)
    if not o_set <= set(src_unique):
        raise ValueError("Output index not present on the input term.")
    if len(set(out_inds)) != len(out_inds):
        raise ValueError("Repeated output indices are not supported.")
    summed = [ix for ix in src_unique if ix not in o_set]
    ext = lambda g: [size_dict[ix] for ix in g]  # noqa: E731
    C = out_ref_factory(tuple(out_inds), tuple(out_inds))

    step = Step(kind=KIND_SINGLE, a=src, c=C, node=node)
    rows_g = list(C.inds)  # iterate outputs in the result's memory order
    step.R = prod(ext(rows_g))
    lo, tabs = _rows_two_level(
        ext(rows_g),
        [
            [src.stride_of(ix) for ix in rows_g],
            [C.stride_of(ix) for ix in rows_g],
        ],
    )
    step.row_lo = lo
    step.rows["A"], step.rows["C"] = tabs
    step.K = prod(ext(summed))
    step.k_lo, (step.k_tabs["A"],) = _rows_two_level(
        ext(summed), [[src.stride_of(ix) for ix in summed]]
    )
    step.macs = 0
    step.elems_rw = step.R * step.K + step.R
    step.label = (
        f"{''.join(map(str, src.inds))}->{''.join(map(str, C.inds))}"
        if len(src.inds) <= 16
        else f"single r{step.R} s{step.K}"
    )
    return step


# --------------------------------------------------------------------------- #
# whole-tree compilation
# --------------------------------------------------------------------------- #


FUSE_MIN_ELEMS = 1 << 24  # stem pairs are fused when the big operand has at least this many elements

# Slice groups (round 4).  The reference contracts every slice from the leaves (core.py:3802-3834); steps that
# depend on no sliced index at all were already computed once (slice-invariant subtrees).  The same holds one
# level down: two slices that differ only in the values of a FEW sliced indices share every step below which
# none of those indices occurs.  With such "group indices" g_1 .. g_k chosen, the executor visits the slices
# group by group (2^k slices that agree on all other sliced indices), computes the shared steps for the first
# slice of a group and keeps what the other steps read of them in memory that per-slice steps do not recycle
# -- 288 GB of HBM hold a 34 GB tensor more than the arena needs.
GROUP_MIN_WIDTH = 1 << 28      # only trees whose slices are launch sequences of large steps
GROUP_MAX_INDS = 3
GROUP_MIN_SAVING = 0.02        # of a slice's modelled time
GROUP_MIN_SAVING_SMALL = 0.15  # ... for trees below GROUP_MIN_WIDTH (slices batched into launches)
GROUP_MAX_KEPT_BYTES = 96 * 2**30
GROUP_MAX_TOTAL_BYTES = 250 * 2**30


def slice_groups_enabled():
    """On unless ``CTG_SLICE_GROUPS`` is "0" / "" (the plan then shares nothing between slices but the
    slice-invariant steps, as in rounds 1-3)."""
    return os.environ.get("CTG_SLICE_GROUPS", "1") not in ("", "0")


def choose_slice_group(tree, plan):
    """Which sliced indices to make the group indices of ``plan`` (compiled for ``tree`` without groups):
    greedily the index whose addition saves most modelled time per slice -- a step that depends on none
    of them costs a slice 2^-k of its time -- while the tensors to be kept for a group fit; ``()`` if
    nothing saves ``GROUP_MIN_SAVING``."""
    if not slice_groups_enabled() or tree.N < 3 or tree.multiplicity < 4:
        return ()
    # trees whose slices go out many per launch (small ones): the executor batches whole groups -- the
    # shared steps once per group of a launch -- which pays only when a good part of a slice is shared,
    # and not at all next to fused stem steps (their kernel does not take part: slice by slice then)
    small = tree.max_size() < GROUP_MIN_WIDTH
    if small and any(s.kind == KIND_STEM2 for s in plan.steps):
        return ()
    min_saving = GROUP_MIN_SAVING_SMALL if small else GROUP_MIN_SAVING
    from .pathfind import step_seconds

    sliced = [si.ind for si in tree.sliced_inds.values() if si.project is None]
    below = {i: frozenset(ix for ix in term if ix in tree.sliced_inds) for i, term in enumerate(tree.inputs)}
    kids = {}
    for p, l, r in tree.traverse():
        below[p] = below[l] | below[r]
        kids[p] = (l, r)
    rows = [(s, step_seconds(s), below.get(s.node, frozenset())) for s in plan.steps if s.node >= 0 and not s.invariant]
    total = sum(t for _, t, _ in rows)
    if total <= 0:
        return ()
    itemsize = plan.itemsize
    chosen, best_saving = [], 0.0
    while len(chosen) < GROUP_MAX_INDS:
        cands = []
        for ix in sliced:
            if ix in chosen:
                continue
            g = set(chosen) | {ix}
            shared = [s for s, _, d in rows if d and not (g & d)]
            # (a shared step costs a slice 1 / group size of itself: the extents of the group indices, not 2 each)
            saving = sum(t for _, t, d in rows if d and not (g & d)) * (1.0 - 1.0 / prod(tree.size_dict[i_] for i_ in g))
            ids = {id(s.c) for s in shared}
            kept = sum(op.size for s, _, d in rows if (g & d)
                       for op in (s.a, s.b, getattr(s, "b2", None), getattr(s, "bm", None))
                       if op is not None and id(op) in ids)
            if kept * itemsize > GROUP_MAX_KEPT_BYTES or (plan.arena_elems + kept) * itemsize > GROUP_MAX_TOTAL_BYTES:
                continue
            cands.append((saving, ix))
        if not cands:
            break
        saving, ix = max(cands, key=lambda c: (c[0], str(c[1])))
        if saving <= best_saving * 1.02:
            break
        chosen.append(ix)
        best_saving = saving
    if best_saving < min_saving * total:
        return ()
    return tuple(chosen)



def compile_tree(tree, dtype, order=None, force_kernel=None, fuse=None, fuse_min_elems=None, _pairs=None,
                 stem_bf16x3=None, _group=None):
    """Compile ``tree`` (possibly sliced) into a :class:`Plan` that computes
    ONE slice and accumulates it into the full result tensor.

    The traversal order and the set of steps are exactly those of the
    reference's ``extract_contractions`` (contract.py:573-651): optional
    per-leaf single-term preprocessing, then one pairwise step per tree
    node, bottom-up.

    ``fuse`` (default: on for complex64 unless ``CTG_NO_FUSE`` is set): pairs of
    consecutive stem steps are emitted as one STEM2 step (stem.py) -- the tree is
    compiled once without, the pairs are chosen on that plan's steps, and the
    tree is compiled again with them.
    """
    if fuse is None:
        fuse = os.environ.get("CTG_NO_FUSE", "0") in ("", "0")
    if _pairs is None and _group is None:
        # top level: compile without fusion, choose the pairs on that plan's steps, compile with them,
        # choose the group indices on THAT plan's steps, compile once more with both
        pairs = {}
        plan = None
        if fuse and dtype == "complex64" and force_kernel is None and tree.N > 2:
            from .stem import find_pairs

            plan = compile_tree(tree, dtype, order, force_kernel, fuse=False, _pairs={}, _group=())
            pairs = find_pairs(
                plan, tree.size_dict,
                min_elems=(
                    int(os.environ.get("CTG_FUSE_MIN_ELEMS", FUSE_MIN_ELEMS))
                    if fuse_min_elems is None else fuse_min_elems
                ),
                bf16x3=stem_bf16x3,   # (the pairs are priced in the arithmetic they will run in)
            )
        if plan is None or pairs:
            plan = compile_tree(tree, dtype, order, force_kernel, fuse=False, _pairs=pairs, _group=())
        group = choose_slice_group(tree, plan)
        if group:
            plan = compile_tree(tree, dtype, order, force_kernel, fuse=False, _pairs=pairs, _group=frozenset(group))
        return plan
    pairs = _pairs or {}
    group = frozenset(_group or ())
    pair_second = {v: k for k, v in pairs.items() if v != k}   # (k: k = a single step on the stem kernel)
    stem_pending = {}  # first node of a pair -> (A, B1, legs of the intermediate)
    plan = Plan(dtype)
    size_dict = tree.size_dict
    N = tree.N
    if N > 1 and tree.max_size() > MAX_TENSOR_ELEMS:
        # (the offset tables alone would not fit the host; the reference lets such
        # a contraction start and die in numpy's allocator)
        raise MemoryError(
            f"the largest intermediate of one slice has 2^{math.log2(tree.max_size()):.1f} elements; "
            f"slice the tree (ContractionTree.slice / pathfind.slice_tree) to at most "
            f"2^{int(math.log2(MAX_TENSOR_ELEMS))} before contracting on one device"
        )

    # -- inputs space: all (unsliced) inputs back to back, 64-element aligned
    cursor = 0
    for term in tree.inputs:
        n = prod(size_dict[ix] for ix in term)
        plan.input_sizes.append(n)
        plan.input_offsets.append(cursor)
        cursor += (n + ARENA_ALIGN - 1) // ARENA_ALIGN * ARENA_ALIGN

    # -- result: the full output tensor (all slices accumulate into it)
    # (an output index projected onto one value keeps a size-1 axis, as the
    # reference's gather_slices stacks over sliced_range = [project], core.py:3866-3876)
    plan.result_shape = tree.gathered_shape()
    plan.result_elems = prod(plan.result_shape)
    full_out_strides = dict(
        zip(tree.output, _row_major_strides(plan.result_shape))
    )

    # -- slicing metadata
    sliced = list(tree.sliced_inds.values())
    plan.nslices = tree.multiplicity
    plan.slice_sizes = [si.size for si in sliced]
    plan.slice_fixed = [(-1 if si.project is None else si.project) for si in sliced]
    strides = np.zeros((N + 1, len(sliced)), dtype=np.int64)
    for i, term in enumerate(tree.inputs):
        shape = [size_dict[ix] for ix in term]
        st = _row_major_strides(shape)
        for j, si in enumerate(sliced):
            strides[i, j] = sum(s for ix, s in zip(term, st) if ix == si.ind)
    for j, si in enumerate(sliced):
        strides[N, j] = 0 if si.project is not None else full_out_strides.get(si.ind, 0)
    plan.slice_strides = strides

    arena = Arena()
    arena_live = {}  # id(TensorRef) -> (offset, nelems)
    # Wide trees keep their small per-slice intermediates in a pool of their own
    # (placed behind the large ones): a few KB allocated first-fit at the start of
    # a 34 GB hole would make that hole useless for the next 34 GB tensor, and the
    # arena a third larger than the two big tensors it has to hold at once.
    small_limit = (tree.max_size() >> 6) if (N > 1 and tree.max_size() >= (1 << 26)) else 0
    sarena = Arena()
    small = set()  # id(TensorRef) of per-slice tensors in the small pool
    small_refs = []

    # Slice-invariant subtrees: a node none of whose
    # leaves is sliced evaluates to the same tensor in every slice; the
    # reference recomputes it per slice (core.py:3821-3823), here it is
    # computed once per upload.  Its output lives in memory that per-slice
    # steps never recycle.  Only meaningful for sliced trees.
    use_invariants = tree.multiplicity > 1
    persistent = set()  # id(TensorRef) of invariant outputs
    persistent_refs = []
    parena = Arena()

    def arena_factory(force_order=None, invariant=False, reorder=None):
        def make(inds, natural):
            order_ = force_order if force_order is not None else natural
            if force_order is None and reorder is not None:
                order_ = reorder(natural)
            shape = [size_dict[ix] for ix in order_]
            n = prod(shape)
            # invariant outputs live in their own region placed behind the
            # per-slice arena (relocated below once its peak is known): they
            # must survive the per-slice recycling of every later slice
            pool = parena if invariant else (sarena if n < small_limit else arena)
            off = pool.alloc(n)
            ref = TensorRef(
                SPACE_ARENA, off, -1, tuple(order_), _row_major_strides(shape), n
            )
            arena_live[id(ref)] = (off, n)
            if invariant:
                persistent.add(id(ref))
                persistent_refs.append(ref)
            elif pool is sarena:
                small.add(id(ref))
                small_refs.append(ref)
            return ref

        return make

    def release(ref):
        if ref.space == SPACE_ARENA and id(ref) not in persistent:
            off, n = arena_live.pop(id(ref))
            (sarena if id(ref) in small else arena).release(off, n)

    producer = {}  # id(TensorRef) -> index of the pair step that wrote it

    cur = {"group": False}

    def add(step):
        step.group = bool(cur["group"]) and not step.invariant
        if step.kind == KIND_PAIR:
            step.a_prod = producer.get(id(step.a), -1)
            step.b_prod = producer.get(id(step.b), -1)
            producer[id(step.c)] = len(plan.steps)
        if step.kind == KIND_STEM2:
            step.a_prod = producer.get(id(step.a), -1)
            step.b_prod = producer.get(id(step.b), -1)
            step.b2_prod = producer.get(id(step.b2), -1) if step.b2 is not None else -1
            if getattr(step, "bm", None) is not None:
                step.bm_prod = producer.get(id(step.bm), -1)
            producer[id(step.c)] = len(plan.steps)
        plan.steps.append(step)
        plan.macs_per_slice += step.macs
        plan.elems_rw_per_slice += step.elems_rw
        plan.elems_moved_per_slice += step.elems_moved if step.kind == KIND_STEM2 else step.elems_rw

    # -- leaves: strided views of the resident inputs
    tensors = {}
    depends = {}  # node -> does it depend on a sliced input?
    # slice groups: does a node depend on one of the group indices?  Known for every node up front,
    # because what matters at a node is its CONSUMER: a result that does not depend on the group but feeds
    # a step that does is kept where per-slice steps never recycle it (like the invariant ones)
    dep_g, parent_of = {}, {}
    if group:
        for i, term in enumerate(tree.inputs):
            dep_g[i] = any(ix in group for ix in term)
        for p_, l_, r_ in tree.traverse():
            dep_g[p_] = dep_g[l_] or dep_g[r_]
            parent_of[l_] = parent_of[r_] = p_
        plan.group_inds = tuple(ix for ix in tree.sliced_inds if ix in group)
        plan.slice_group = [1 if si.ind in group else 0 for si in tree.sliced_inds.values()]

    def last_of(node):   # the node whose step consumes / emits a fused chain starting at ``node``
        while node in pairs and pairs[node] != node:
            node = pairs[node]
        return node

    def kept_for_group(node):
        """``node``'s result is per-group work consumed by per-slice work."""
        if not group or dep_g[node] or not depends[node] or node not in parent_of:
            return False
        return dep_g[last_of(parent_of[node])]
    # LDS-resident subtrees (round 6, ldsrun.py): maximal subtrees whose tensors all fit one compute
    # unit's LDS are run by one workgroup of one launch.  Their ordinary steps stay in the plan -- complete,
    # FIRST in the step order among the steps of their sharing class (the arena is assigned for the order the
    # launches really have) -- next to a second lowering on LDS-resident tensors (Plan.lds_runs).
    comps = []
    node_class = None
    leaf_order = range(N)
    if (order is None and N > 3 and tree.max_size() <= LEVEL_ORDER_MAX_ELEMS and force_kernel is None):
        from . import ldsrun

        if ldsrun.lds_runs_enabled():
            for i in range(N):
                tree.get_legs(i)   # (fills tree.preprocessing)
            dep_s = dict((i, (i in tree.sliced_inputs) or not use_invariants) for i in range(N))
            for p_, l_, r_ in tree.traverse():
                dep_s[p_] = dep_s[l_] or dep_s[r_] or p_ == tree.root

            def node_class(node):
                # (a leaf: the class of its preprocessing step)
                if not dep_s[node]:
                    return "inv"
                if group and not dep_g[node] and node != tree.root:
                    return "group"
                return "slice"

            comps = ldsrun.choose_components(tree, dtype, plan.itemsize, node_class, pairs=pairs)
        if comps:
            # (leaf preprocessing steps in the same class order as the pair steps below)
            leaf_order = sorted(range(N), key=lambda i: ldsrun.CLASS_RANK[node_class(i)])
    for i in leaf_order:
        term = tree.inputs[i]
        full_shape = [size_dict[ix] for ix in term]
        st = _row_major_strides(full_shape)
        kept = [(ix, s) for ix, s in zip(term, st) if ix not in tree.sliced_inds]
        view = TensorRef(
            SPACE_INPUTS,
            plan.input_offsets[i],
            i,
            tuple(ix for ix, _ in kept),
            tuple(s for _, s in kept),
            plan.input_sizes[i],
        )
        legs = tree.get_legs(i)  # also fills tree.preprocessing lazily
        depends[i] = (i in tree.sliced_inputs) or not use_invariants
        if i in tree.preprocessing and N > 1:
            inv = not depends[i]
            cur["group"] = bool(group) and depends[i] and not dep_g[i]
            step = build_single_step(
                size_dict, view, tuple(legs), arena_factory(invariant=inv or kept_for_group(i)), node=i
            )
            step.invariant = inv
            add(step)
            tensors[i] = step.c
        else:
            tensors[i] = view

    root_order = tuple(ix for ix in tree.output if ix not in tree.sliced_inds)

    # Small trees are executed wave front by wave front: a slice of such a tree is
    # a chain of launches of a few microseconds each, so the steps are emitted
    # level by level (level = 1 + the deeper child; the steps of a level are
    # independent), cheapest first, and operands are recycled only once their
    # whole level is done -- the executor then sends consecutive independent
    # small steps out as one launch (ctg_runtime.hip: build_groups).  The
    # contracted values do not depend on the order, only lifetimes do; wide
    # trees keep the reference's depth-first order and its smaller peak memory.
    level = None
    if order is None and N > 3 and tree.max_size() <= LEVEL_ORDER_MAX_ELEMS:
        level = {}
        for p, l, r in tree.traverse():
            level[p] = 1 + max(level.get(l, 0), level.get(r, 0))
        order = lambda node: (level[node], tree.get_flops(node))  # noqa: E731

    # Consumer-aware layouts (experiment, CTG_PLAN_CONSUMER_ORDER=1, small trees only): an
    # intermediate keeps the indices its parent step keeps first and the ones it contracts
    # last, in one canonical order -- both operands of the parent step are then contiguous
    # along k.  Any order is legal (every kernel gathers through tables); the default is
    # the reference's [kept-left..., kept-right...] (core.py:1035-1051).
    consumer_order = None
    if level is not None and os.environ.get("CTG_PLAN_CONSUMER_ORDER", "0") not in ("", "0"):
        parent_of = {}
        for p, l, r in tree.traverse():
            parent_of[l] = parent_of[r] = p

        def consumer_order(node):
            up = parent_of.get(node)
            if up is None:
                return None
            kept_up = set(root_order) if up == tree.root else set(tree.get_legs(up))

            def reorder(natural):
                return tuple(ix for ix in natural if ix in kept_up) + tuple(
                    sorted((ix for ix in natural if ix not in kept_up), key=str)
                )

            return reorder

    sequence = None
    if comps and level is not None:
        rank = ldsrun.CLASS_RANK
        member_nodes = set(n for c in comps for n in c.nodes)
        seq = list(tree.traverse(order=order))
        # What members read from the arena -- results of a class that runs less often -- comes before
        # them: class by class (each is closed under "child of"), members of a class before its other steps.
        sequence = sorted(seq, key=lambda x: (rank[node_class(x[0])], x[0] not in member_nodes))
    operand_node = {}   # id(TensorRef) -> the tree node whose tensor it is (LDS shadows)
    step_of_node = {}

    if N == 1:
        step = build_single_step(
            size_dict, tensors[0], root_order, arena_factory(), node=tree.root
        )
        add(step)
        final = step.c
    else:
        final = None
        pending, cur_level = [], 0
        for i_, ref_ in tensors.items():
            operand_node[id(ref_)] = i_
        for p, l, r in (sequence if sequence is not None else tree.traverse(order=order)):
            if level is not None and level[p] != cur_level:
                for ref in pending:
                    release(ref)
                pending, cur_level = [], level[p]
            is_root = p == tree.root
            depends[p] = depends[l] or depends[r] or is_root
            inv = not depends[p]
            cur["group"] = bool(group) and depends[p] and not dep_g[p] and not is_root
            factory = arena_factory(
                root_order if is_root else None,
                invariant=inv or kept_for_group(p),
                reorder=consumer_order(p) if consumer_order is not None else None,
            )
            p_inds = root_order if is_root else tuple(tree.get_legs(p))
            tl, tr = tensors.pop(l), tensors.pop(r)
            steps_new = None
            if pairs.get(p) == p:
                # a large step no pair took: the stem kernel's first half alone (stem.build_stem_one)
                from .stem import build_stem_one

                A1, B1 = pick_rows_operand(size_dict, tl, tr, p_inds)
                one = build_stem_one(size_dict, A1, B1, p_inds, factory, node=p)
                if one is not None:
                    steps_new = [one]
            elif p in pairs and p in pair_second and not is_root and (
                    id(tl) in stem_pending or id(tr) in stem_pending):
                # middle step of a three-step tile (stem.build_stem_triple): still nothing emitted;
                # the chain (A, B1, first node, BM, legs of the first intermediate) waits for the last
                virt, other = (tl, tr) if id(tl) in stem_pending else (tr, tl)
                chain = stem_pending.pop(id(virt))
                if len(chain) == 3 and pick_rows_operand(size_dict, virt, other, p_inds)[0] is virt:
                    virt2 = TensorRef(-2, 0, -1, p_inds, (0,) * len(p_inds), prod(size_dict[ix] for ix in p_inds))
                    stem_pending[id(virt2)] = chain + (other, virt.inds, p)
                    tensors[p] = virt2
                    continue
                # (does not chain after all: the first step as usual, this one starts a pair)
                A1, B1, p1 = chain[:3]
                first = build_pair_step(dtype, size_dict, A1, B1, virt.inds, arena_factory(invariant=inv), node=p1)
                first.invariant = inv
                add(first)
                for ref in (first.a, first.b):
                    release(ref) if level is None else pending.append(ref)
                if virt is tl:
                    tl = first.c
                else:
                    tr = first.c
                A1, B1 = pick_rows_operand(size_dict, tl, tr, p_inds)
                virt = TensorRef(-2, 0, -1, p_inds, (0,) * len(p_inds), prod(size_dict[ix] for ix in p_inds))
                stem_pending[id(virt)] = (A1, B1, p)
                tensors[p] = virt
                continue
            elif p in pairs and not is_root:
                # first step of a fused pair: nothing is emitted, nothing allocated; the
                # operands stay alive until the second step takes them
                A1, B1 = pick_rows_operand(size_dict, tl, tr, p_inds)
                virt = TensorRef(-2, 0, -1, p_inds, (0,) * len(p_inds), prod(size_dict[ix] for ix in p_inds))
                stem_pending[id(virt)] = (A1, B1, p)
                tensors[p] = virt
                continue
            if p in pair_second:
                from .stem import build_stem_step

                virt, other = (tl, tr) if id(tl) in stem_pending else (tr, tl)
                chain = stem_pending.pop(id(virt))
                if len(chain) == 6:
                    # last step of a three-step tile
                    from .stem import build_stem_triple

                    A1, B1, p1, BM, c1_inds, pm = chain
                    triple = None
                    if pick_rows_operand(size_dict, virt, other, p_inds)[0] is virt:
                        triple = build_stem_triple(size_dict, A1, B1, BM, other, c1_inds, virt.inds, p_inds, factory, node=p)
                    if triple is not None:
                        triple.invariant = inv
                        add(triple)
                        for ref in (triple.a, triple.b, triple.bm, triple.b2):
                            release(ref) if level is None else pending.append(ref)
                        tensors[p] = triple.c
                        final = triple.c
                        continue
                    # (no tile after all: the first two as a pair -- or one by one -- then this step)
                    two = build_stem_step(size_dict, A1, B1, BM, c1_inds, virt.inds, arena_factory(invariant=inv), node=pm)
                    if two is None:
                        one_ = build_pair_step(dtype, size_dict, A1, B1, c1_inds, arena_factory(invariant=inv), node=p1)
                        two_ = build_pair_step(dtype, size_dict, one_.c, BM, virt.inds, arena_factory(invariant=inv), node=pm)
                        emitted = [one_, two_]
                    else:
                        emitted = [two]
                    for st_ in emitted:
                        st_.invariant = inv
                        add(st_)
                        ops_ = [st_.a, st_.b] + ([st_.b2] if st_.kind == KIND_STEM2 and st_.b2 is not None else [])
                        for ref in ops_:
                            release(ref) if level is None else pending.append(ref)
                    if virt is tl:
                        tl = emitted[-1].c
                    else:
                        tr = emitted[-1].c
                    chain = None
                fused = None
                if chain is not None:
                    A1, B1, p1 = chain
                if chain is not None and pick_rows_operand(size_dict, virt, other, p_inds)[0] is virt:
                    fused = build_stem_step(size_dict, A1, B1, other, virt.inds, p_inds, factory, node=p)
                if fused is not None:
                    steps_new = [fused]
                elif chain is not None:
                    # the pair does not fit after all: both steps as usual, one after the other
                    first = build_pair_step(
                        dtype, size_dict, A1, B1, virt.inds, arena_factory(invariant=inv), node=p1,
                    )
                    if virt is tl:
                        tl = first.c
                    else:
                        tr = first.c
                    steps_new = [first]
            if steps_new is None or steps_new[0].kind == KIND_PAIR:
                steps_new = (steps_new or []) + [build_pair_step(
                    dtype,
                    size_dict,
                    tl,
                    tr,
                    p_inds,
                    factory,
                    node=p,
                    force_kernel=force_kernel,
                )]
            for step in steps_new:
                step.invariant = inv
                add(step)
                operands = [step.a, step.b] + ([step.b2] if step.kind == KIND_STEM2 and step.b2 is not None else [])
                if level is not None:
                    pending += operands
                else:
                    for ref in operands:
                        release(ref)
            step = steps_new[-1]
            tensors[p] = step.c
            operand_node[id(step.c)] = p
            step_of_node[p] = step
            final = step.c
        for ref in pending:
            release(ref)

    # -- accumulate the slice into the full result at its chunk position
    acc = Step(kind=KIND_ACCUM, a=final, node=-1, label="accumulate")
    res = TensorRef(
        SPACE_RESULT,
        0,
        N,  # pseudo-leaf N carries the chunk offset of outer-sliced indices
        root_order,
        tuple(full_out_strides[ix] for ix in root_order),
        plan.result_elems,
    )
    acc.c = res
    rows_g = list(final.inds)
    ext = [size_dict[ix] for ix in rows_g]
    acc.R = prod(ext)
    lo, tabs = _rows_two_level(
        ext,
        [
            [final.stride_of(ix) for ix in rows_g],
            [res.stride_of(ix) for ix in rows_g],
        ],
    )
    acc.row_lo = lo
    acc.rows["A"], acc.rows["C"] = tabs
    acc.elems_rw = 0
    add(acc)
    release(final)

    if comps:
        from . import ldsrun

        ldsrun.build_shadows(plan, tree, dtype, comps, step_of_node, operand_node)

    big_peak = max(arena.peak, ARENA_ALIGN)
    for ref in small_refs:
        ref.offset += big_peak
    slice_peak = big_peak + sarena.peak
    for ref in persistent_refs:
        ref.offset += slice_peak
    plan.arena_elems = slice_peak + parena.peak
    plan.inputs_elems = max(cursor, ARENA_ALIGN)
    return plan
