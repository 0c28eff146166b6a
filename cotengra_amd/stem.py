"""Fused stem pairs: two consecutive pairwise steps of a contraction stem as ONE
launch whose intermediate never touches HBM.

A sliced Sycamore tree spends its time on a *stem*: a 2^32-element tensor to
which small tensors (a few hundred to a few thousand elements) are applied one
after the other, each step contracting 4-7 binary indices of the big tensor and
putting 4-7 new ones in their place.  Executed step by step (the reference's
loop, ``cotengra/contract.py:788-832``, one ``tensordot`` per node) every step
reads the big tensor and writes it back: 4 x 34 GB for two steps.  Here two
consecutive steps

    C1[r1, n1] = sum_k1  A[r1, k1] B1[k1, n1]
    C2[r2, n2] = sum_k2 C1[r2, k2] B2[k2, n2]        (r2 u k2 = r1 u n1)

run over *tiles*: the index bits of ``A`` are split into tile bits -- all of
``k1``, the part of ``k2`` that lives on ``A`` (``k2r``), and the lowest-stride
bits of ``A`` that fill a tile up to 256 rows (``X``) -- and grid bits ``G``.
One workgroup (8 waves) takes one value of ``G`` at a time: every wave gathers
32 rows x K1 of its tile from HBM straight into matrix-core fragments (a lane =
one row and one of the two k-rows of the instruction) and multiplies by ``B1``; the 256 x N1 result goes to LDS laid
out as the second step's operand ``[r2][k2]``; the second step reads its
fragments from there and stores ``C2``.  HBM sees ``A`` once and ``C2`` once.

A large step no pair took runs on the same kernel's first half alone (round 4:
``geometry_one`` / ``build_stem_one``); three consecutive steps as one tile exist
behind ``CTG_STEM_TRIPLES`` (``geometry3`` / ``build_stem_triple``: parity green,
measured slower than pair + single step on every m20 tree -- DESIGN.md section 8).

This module is host-side planning only: which pairs to fuse (``find_pairs``)
and the offset tables of a fused step (``build_stem_step``); the kernel is
``csrc/ctg_stem.hip``, the numpy restatement of its addressing
``oracle/plan_interp.py`` (test infrastructure).  Everything here is index
work on powers of two: every extent is split into binary digits ("bits") and
every table is additive over bits, which is what lets the kernel form an
address as *uniform part + per-lane constant*.
"""

from __future__ import annotations

import math
import os

import numpy as np

from . import plan as P

WAVES = 8                    # waves of a workgroup = row tiles of step 1 in flight
LDS_BYTES = 160 * 1024       # per CU (one workgroup per CU)
LDS_SLACK = 256
G_LO_BITS = 12               # fast level of the two-level grid tables

DESC_WORDS = 56              # header of the serialised descriptor (int64 words; 40 until ABI 4)
DESC_MAGIC = 0x53544D33      # "STM3"

# what the kernel is instantiated for (csrc/ctg_stem.hip: launch_stem2)
K_OK = (16, 32, 64, 128)
N1_OK = (16, 32, 64, 128)
N2_OK = (16, 32, 64, 128)

# Time model of a fused pair, from knock-out builds of the kernel on the m20 stem
# (tools/exp_stem_ko.sh, profiles/r3_stem_knockout.txt): without its memory traffic the
# kernel runs its matrix work at 0.73 of the 157.3 TFLOP/s peak; without its MFMAs it
# moves A at a rate set by how many bytes of a wave's 32 x 16 task are contiguous in
# memory (a tile that has no room for A's lowest-stride digits uses 32-byte pieces of
# the lines it fetches), C2 at the full rate; together they take the longer of the two plus a fifth
# of the shorter.
FUSED_MFMA_RATE = 157.3e12 * 0.73
# ... and in the bf16 x 3 arithmetic (csrc/ctg_stem.hip: BF3; fp32 operands split into three bf16
# values, products on the bf16 matrix cores -- the default since round 4, ``bf16x3_mode``): the
# factor by which the pairs' matrix work speeds up, as measured on whole pairs (DESIGN.md section
# 4b); tree refinement for that mode (tests/golden/gen/refine_bf3.py) prices pairs with it
BF16X3_SPEEDUP = 1.6
FUSED_STORE_RATE = 5.4e12
FUSED_OVERLAP_LOSS = 0.2
MIN_GAIN = 0.05              # fuse only if the model saves at least this fraction


def bf16x3_mode(flag=None):
    """Arithmetic of the fused stem pairs.  Since round 4 the default is
    bf16 x 3: fp32 operands split exactly into three bfloat16 limbs, six products on the bf16
    matrix cores, fp32 accumulation -- the fp32 kernel's accuracy on every adversarial test,
    11-13 % less time per slice.  ``CTG_STEM_BF16X3`` in the environment decides when it is set
    ("0" / "" = fp32 products on the fp32 matrix cores, anything else = bf16 x 3; the C side
    reads it at every launch the same way), else the caller's ``flag``
    (``HipContractor(stem_bf16x3=...)``, ``ctg_exec_set_stem_arithmetic``), else the default."""
    v = os.environ.get("CTG_STEM_BF16X3")
    if v is not None:
        return v not in ("", "0")
    return True if flag is None else bool(flag)


def gather_rate(run_bytes):
    """Bytes per second of the A gather by contiguous run length."""
    if run_bytes >= 256:
        return 5.4e12
    if run_bytes >= 128:
        return 4.8e12
    if run_bytes >= 64:
        return 2.85e12
    return 1.25e12


def _log2(n):
    n = int(n)
    if n < 1 or n & (n - 1):
        return None
    return n.bit_length() - 1


def _bits_of(inds, size_dict):
    """Binary digits of an index list, least significant digit of each index
    first: ``(ix, j)`` stands for the factor ``2^j`` of index ``ix``."""
    out = []
    for ix in inds:
        b = _log2(size_dict[ix])
        if b is None:
            return None
        out += [(ix, j) for j in range(b)]
    return out


def _stride(ref, bit):
    ix, j = bit
    return ref.stride_of(ix) << j


def _table(bits, strides):
    """Offset table over an index whose binary digit ``p`` is ``bits[p]``
    (position 0 = least significant): entry ``i`` = sum of the strides of the
    digits set in ``i``."""
    t = np.zeros(1, dtype=np.int64)
    for s in strides:
        t = np.concatenate([t, t + int(s)])
    return t


def b_lds_bytes(K, N):
    """LDS of a small operand's planes: (Re, Im) -- plus -Im when the 16 output
    columns share one matrix-core tile with their own imaginary parts."""
    return (3 if N == 16 else 2) * N * (K + 4) * 4


def bf16x3_fits(K1, N1, K2, N2, rows2):
    """Does the pair's tile fit the LDS in the bf16 x 3 arithmetic (the small operands as three
    bfloat16 limb planes: csrc/ctg_stem.hip ``stem2_lds_bytes_bf3``)?  A pair that does not runs
    fp32 products whatever the mode (``stem2_bf3``), and is priced so."""
    q1 = (3 if N1 == 16 else 2) * N1 * ((K1 >> 4) * 48 + 8)
    q2 = (3 if N2 == 16 else 2) * N2 * ((K2 >> 3) * 24 + 8)
    return 2 * (q1 + q2) + 8 * rows2 * (K2 + 4) + 8 * N2 + 64 <= LDS_BYTES


class Geometry:
    """Tile decomposition of one fused pair (all fields are plain data)."""


def _classify(step, size_dict):
    """Index groups of a plain matrix-core pair step, or None if the step has
    anything a fused pair does not handle (batch indices, indices summed on one
    operand only, extents that are not powers of two)."""
    if step.kind != P.KIND_PAIR or step.kernel != P.KERNEL_MFMA or step.Bt != 1:
        return None
    a, b, c = step.a, step.b, step.c
    for t in (a, b, c):
        if len(set(t.inds)) != len(t.inds):
            return None
    a_set, b_set, o_set = set(a.inds), set(b.inds), set(c.inds)
    if a_set & b_set & o_set:
        return None
    con = [ix for ix in a.inds if ix not in o_set]
    if any(ix not in b_set for ix in con):
        return None
    if any(ix not in o_set and ix not in a_set for ix in b.inds):
        return None
    keep_b = [ix for ix in b.inds if ix in o_set]
    keep_a = [ix for ix in a.inds if ix in o_set]
    if set(keep_a) | set(keep_b) != o_set:
        return None
    groups = {"con": con, "keep_a": keep_a, "keep_b": keep_b}
    for g in groups.values():
        if _bits_of(g, size_dict) is None:
            return None
    return groups


def geometry(size_dict, A, B1, B2, c1_inds, c2_inds):
    """Tile decomposition of the pair ``(A, B1 -> c1_inds)``, ``(C1, B2 ->
    c2_inds)``; ``None`` if the pair does not fit the kernel."""
    o1, o2 = set(c1_inds), set(c2_inds)
    a_bits = _bits_of(A.inds, size_dict)
    k1 = _bits_of([ix for ix in A.inds if ix not in o1], size_dict)
    n1 = _bits_of([ix for ix in B1.inds if ix in o1], size_dict)
    k2 = _bits_of([ix for ix in c1_inds if ix not in o2], size_dict)
    n2 = _bits_of([ix for ix in B2.inds if ix in o2], size_dict)
    if None in (a_bits, k1, n1, k2, n2):
        return None
    K1, N1, K2, N2 = (1 << len(g) for g in (k1, n1, k2, n2))
    if K1 not in K_OK or K2 not in K_OK or N1 not in N1_OK or N2 not in N2_OK:
        return None
    k1_set, n1_set, k2_set = set(k1), set(n1), set(k2)
    r1 = [b for b in a_bits if b not in k1_set]
    k2r = [b for b in k2 if b not in n1_set]
    k2n = [b for b in k2 if b in n1_set]
    if any(b not in set(r1) for b in k2r):
        return None

    def sa(b):
        return _stride(A, b)

    free = sorted((b for b in r1 if b not in k2_set), key=sa)   # candidates for X, lowest stride first
    best = None
    cs1 = max(1, N1 // 32)   # 32-column groups of step 1: a unit = (32-row tile, column group)
    for units in (WAVES, 2 * WAVES):
        nr1 = 5 + _log2(units // cs1)
        nx = nr1 - len(k2r)
        if nx < 0 or nx > len(free):
            continue
        tm = nr1 + len(n1)
        rows2_bits = tm - len(k2)
        if rows2_bits < 5:
            continue
        rows2 = 1 << rows2_bits
        mid_bytes = 2 * rows2 * (K2 + 4) * 4
        lds = mid_bytes + b_lds_bytes(K1, N1) + b_lds_bytes(K2, N2) + 8 * N2 + LDS_SLACK
        if lds > LDS_BYTES:
            continue
        ng2 = max(1, N2 // 32)
        items = (rows2 // 32) * ng2
        # a 16-deep first contraction is ONE 4 KB task per wave and tile: twice the rows
        # (two tasks per wave in flight) keep the memory system busier -- 4 KB per wave are
        # 32 KB per CU, below what the latency-bandwidth product of HBM asks for
        deep = 1 if (K1 == 16 and units == 2 * WAVES) else 0
        cand = (min(items, WAVES), deep, -nr1, nr1, nx, rows2_bits, ng2, items, lds)
        if best is None or cand > best:
            best = cand
    if best is None:
        return None
    _, _, _, nr1, nx, rows2_bits, ng2, items, lds = best
    g = Geometry()
    g.K1, g.N1, g.K2, g.N2 = K1, N1, K2, N2
    g.nr1, g.rows2_bits, g.ng2, g.items, g.lds = nr1, rows2_bits, ng2, items, lds
    # (the bf16 x 3 instantiations are static ones: item counts that are multiples of the waves)
    g.bf3_fits = bf16x3_fits(K1, N1, K2, N2, 1 << rows2_bits) and items % WAVES == 0
    g.k1 = sorted(k1, key=sa)                      # k index: digit 0 = lowest stride in A
    g.n1 = sorted(n1, key=lambda b: _stride(B1, b))
    x = free[:nx]
    g.r1 = sorted(k2r + x, key=sa)                 # tile rows of step 1, digit 0 = lowest stride in A
    r1_set = set(g.r1)
    g.grid = sorted((b for b in r1 if b not in r1_set), key=sa)
    g.k2 = k2n + sorted(k2r, key=sa)               # k2 index: the fresh columns first
    g.r2_members = x + [b for b in g.n1 if b not in k2_set]
    g.n2 = n2
    # what one wave gathers per task = 32 rows x 16 k, straight into matrix-core fragments:
    # lane l = (row l & 31, k-row h = l >> 5) loads the 8 elements ("slots") of its row whose
    # k has k-row h, at  task base + kj_a[slot] + lane_a[l]  (the lane part as a 32-bit byte
    # offset).  Which bit of k is the k-row: bit 0 -- slot s is k = 2 s + h, eight 8-byte
    # loads -- unless A's stride-1 digit is a contracted one: then a lane takes k, k + 1 in
    # one 16-byte load, the k-row is bit 1 and slot s is k = 4 (s >> 1) + 2 h + (s & 1).
    g.row_a = _table(g.r1, [sa(b) for b in g.r1])              # [2^nr1] tile rows of A
    g.k_a = _table(g.k1, [sa(b) for b in g.k1])                # [K1]
    g.vec = bool(g.k_a[1] == 1 and A.offset % 2 == 0)
    hbit = 2 if g.vec else 1
    if A.leaf >= 0 or 8 * int(g.row_a[31] + g.k_a[hbit]) >= 1 << 32:
        return None
    # bytes of A that are contiguous in memory within one task (32 rows x 16 k): how much of
    # every 128-byte line a wave fetches it uses itself.  (The 8 load instructions of a task
    # each cover 32 rows x 2 k-rows, i.e. possibly shorter pieces: measured, that does not
    # matter -- the pieces of a line are requested within a few hundred cycles of each other.)
    task = np.sort((g.row_a[:32, None] + g.k_a[None, :16]).reshape(-1))
    runs = np.flatnonzero(np.diff(task) != 1)
    g.run_bytes = 8 * int(runs[0] + 1 if len(runs) else len(task))
    return g


def geometry_one(size_dict, A, B1, c_inds):
    """Tile decomposition of ONE stem step  C[r, n] = sum_k A[r, k] B1[k, n]  run by the stem
    kernel's first half alone (gather straight into matrix-core fragments -> MFMA -> stores from
    the accumulators; no LDS for the big tensors, no barrier): the step a chain of odd length
    leaves over.  ``None`` if the step does not fit (K in 16..128, N in 32..128, 256 or 512 tile
    rows out of A's lowest-stride free digits)."""
    o1 = set(c_inds)
    a_bits = _bits_of(A.inds, size_dict)
    k1 = _bits_of([ix for ix in A.inds if ix not in o1], size_dict)
    n1 = _bits_of([ix for ix in B1.inds if ix in o1], size_dict)
    if None in (a_bits, k1, n1):
        return None
    K1, N1 = 1 << len(k1), 1 << len(n1)
    if K1 not in K_OK or N1 not in (32, 64, 128):
        return None
    k1_set = set(k1)

    def sa(b):
        return _stride(A, b)

    free = sorted((b for b in a_bits if b not in k1_set), key=sa)
    cs1 = N1 // 32
    lds = b_lds_bytes(K1, N1) + LDS_SLACK
    if lds > LDS_BYTES:
        return None
    # (a 16-deep contraction is ONE 4 KB task per unit: two units per wave keep two in flight)
    units = 2 * WAVES if K1 == 16 else WAVES
    nr1 = 5 + _log2(units // cs1) if units >= cs1 else None
    if nr1 is None or nr1 < 5 or nr1 > len(free):
        return None
    g = Geometry()
    g.K1, g.N1, g.K2, g.N2 = K1, N1, 0, 0
    g.nr1, g.rows2_bits, g.ng2, g.items, g.lds = nr1, 0, 0, 0, lds
    # (B1's limb planes: csrc/ctg_stem.hip stem2_lds_bytes_one; K = N = 128 has no room and multiplies in fp32)
    g.bf3_fits = 2 * 2 * N1 * ((K1 >> 4) * 48 + 8) + 8 * N1 + 64 <= LDS_BYTES
    g.k1 = sorted(k1, key=sa)
    g.n1 = sorted(n1, key=lambda b: _stride(B1, b))
    g.r1 = free[:nr1]
    r1_set = set(g.r1)
    g.grid = sorted((b for b in free if b not in r1_set), key=sa)
    g.row_a = _table(g.r1, [sa(b) for b in g.r1])
    g.k_a = _table(g.k1, [sa(b) for b in g.k1])
    g.vec = bool(g.k_a[1] == 1 and A.offset % 2 == 0)
    hbit = 2 if g.vec else 1
    if A.leaf >= 0 or 8 * int(g.row_a[31] + g.k_a[hbit]) >= 1 << 32:
        return None
    task = np.sort((g.row_a[:32, None] + g.k_a[None, :16]).reshape(-1))
    runs = np.flatnonzero(np.diff(task) != 1)
    g.run_bytes = 8 * int(runs[0] + 1 if len(runs) else len(task))
    return g


def single_seconds(macs, elems_a, elems_c, run_bytes=256, bf16x3=None, bf3_fits=True):
    """Modelled time of a single stem step (``geometry_one``): the pair model with one step."""
    rate = FUSED_MFMA_RATE * (BF16X3_SPEEDUP if (bf3_fits and bf16x3_mode(bf16x3)) else 1.0)
    t_mfma = 8.0 * macs / rate
    t_mem = 8.0 * elems_a / gather_rate(run_bytes) + 8.0 * elems_c / FUSED_STORE_RATE
    return max(t_mfma, t_mem) + FUSED_OVERLAP_LOSS * min(t_mfma, t_mem)


def pair_seconds(macs1, macs2, elems_a, elems_c2, items, run_bytes=256, bf16x3=None, bf3_fits=True):
    """Modelled time of a fused pair in the arithmetic ``bf16x3_mode(bf16x3)`` says
    (``bf3_fits``: ``Geometry.bf3_fits`` -- a tile too large for the limb planes multiplies in fp32)."""
    rate = FUSED_MFMA_RATE * (BF16X3_SPEEDUP if (bf3_fits and bf16x3_mode(bf16x3)) else 1.0)
    t_mfma = 8.0 * macs1 / rate + 8.0 * macs2 / (rate * min(1.0, items / WAVES))
    t_mem = 8.0 * elems_a / gather_rate(run_bytes) + 8.0 * elems_c2 / FUSED_STORE_RATE
    return max(t_mfma, t_mem) + FUSED_OVERLAP_LOSS * min(t_mfma, t_mem)


def find_pairs(plan, size_dict, min_elems=1 << 24, model=None, bf16x3=None):
    """Which consecutive stem steps of ``plan`` (compiled without fusion) to fuse:
    ``{node of the first step: node of the second}``.  Candidates are pairs
    (s1, s2) where s2's row operand is s1's result, both plain matrix-core steps
    over binary indices with shapes the kernel takes; a chain of candidates is
    paired off by dynamic programming on the modelled time saved.  Round 4: a large
    step that ends up in no pair (chains of odd length, partners that pair better
    elsewhere) may run on the stem kernel's first half alone -- ``{node: node}``, a
    "pair" with itself (``geometry_one`` / ``build_stem_one``) -- when the model says
    that beats the tiled kernel (it does in the bf16 x 3 arithmetic, where the step
    becomes memory-bound; in fp32 arithmetic the two are equal and nothing is chosen)."""
    if plan.dtype != "complex64":
        return {}
    if model is None:
        from .pathfind import MI355X_C64 as model
    steps = plan.steps
    by_out = {id(s.c): i for i, s in enumerate(steps) if s.kind == P.KIND_PAIR}
    cls = {}

    def classify(i):
        if i not in cls:
            cls[i] = _classify(steps[i], size_dict)
        return cls[i]

    def unfused_seconds(s):
        return model.step_seconds(s.macs, s.elems_rw, s.K, s.N)

    gain = {}   # i2 -> (i1, seconds saved)
    for i2, s2 in enumerate(steps):
        i1 = by_out.get(id(s2.a)) if s2.kind == P.KIND_PAIR else None
        if i1 is None:
            continue
        s1 = steps[i1]
        if s1.invariant != s2.invariant or s1.a.size < min_elems or s1.a.leaf >= 0:
            continue
        if classify(i1) is None or classify(i2) is None:
            continue
        geo = geometry(size_dict, s1.a, s1.b, s2.b, s1.c.inds, s2.c.inds)
        if geo is None:
            continue
        before = unfused_seconds(s1) + unfused_seconds(s2)
        after = pair_seconds(s1.macs, s2.macs, s1.a.size, s2.c.size, geo.items, geo.run_bytes, bf16x3=bf16x3,
                             bf3_fits=geo.bf3_fits)
        if before - after >= MIN_GAIN * before:
            gain[i2] = (i1, before - after)
    if triples_enabled() and bf16x3_mode(bf16x3):
        return _find_chains(steps, size_dict, by_out, classify, unfused_seconds, gain, min_elems, bf16x3)
    # chains: i1 -> i2 -> i3 ...; a step can be in one pair only
    best = {}   # step -> (total gain of the chain ending here, pairs chosen)
    order = sorted(gain)
    for i2 in order:
        i1, g12 = gain[i2]
        skip = best.get(i1, (0.0, ()))                      # i1 free or paired backwards
        before_i1 = best.get(gain[i1][0], (0.0, ())) if i1 in gain else (0.0, ())
        take = (before_i1[0] + g12, before_i1[1] + ((i1, i2),))
        best[i2] = take if take[0] > skip[0] else skip
    # collect: walk every chain from its last step
    chosen = {}
    used = set()
    for i2 in sorted(best, reverse=True):
        if i2 in used:
            continue
        for a, b in best[i2][1]:
            if a not in used and b not in used:
                chosen[steps[a].node] = steps[b].node
                used.update((a, b))
        # everything on this chain is settled
        j = i2
        while j in gain:
            used.add(j)
            j = gain[j][0]
        used.add(j)
    # large steps left alone: the stem kernel's first half, where the model prefers it
    in_pair = {n for kv in chosen.items() for n in kv}
    if os.environ.get("CTG_NO_STEM_ONE", "0") in ("", "0"):
        for i, s1 in enumerate(steps):
            if s1.kind != P.KIND_PAIR or s1.node in in_pair or s1.a.size < min_elems or s1.a.leaf >= 0:
                continue
            if classify(i) is None:
                continue
            geo = geometry_one(size_dict, s1.a, s1.b, s1.c.inds)
            if geo is None:
                continue
            before = unfused_seconds(s1)
            after = single_seconds(s1.macs, s1.a.size, s1.c.size, geo.run_bytes, bf16x3=bf16x3, bf3_fits=geo.bf3_fits)
            if before - after >= MIN_GAIN * before:
                chosen[s1.node] = s1.node
    return chosen


def _find_chains(steps, size_dict, by_out, classify, unfused_seconds, gain2, min_elems, bf16x3):
    """``find_pairs`` with three-step tiles among the choices (``CTG_STEM_TRIPLES``): a dynamic
    programme along every chain of stem steps over "alone", "second of a pair" and "last of a
    triple".  A triple ``n1 -> n2 -> n3`` comes back as the two links ``{n1: n2, n2: n3}``."""
    def ok(i):
        s = steps[i]
        return s.kind == P.KIND_PAIR and classify(i) is not None

    pred = {}
    for i, s in enumerate(steps):
        j = by_out.get(id(s.a)) if s.kind == P.KIND_PAIR else None
        if j is not None and ok(i) and ok(j) and steps[j].invariant == s.invariant:
            pred[i] = j
    gain3 = {}
    for i3, i2 in pred.items():
        i1 = pred.get(i2)
        if i1 is None:
            continue
        s1, s2, s3 = steps[i1], steps[i2], steps[i3]
        if s1.a.size < min_elems or s1.a.leaf >= 0:
            continue
        geo = geometry3(size_dict, s1.a, s1.b, s2.b, s3.b, s1.c.inds, s2.c.inds, s3.c.inds)
        if geo is None or not _triple_instantiated(geo):
            continue
        before = unfused_seconds(s1) + unfused_seconds(s2) + unfused_seconds(s3)
        after = triple_seconds((s1.macs, s2.macs, s3.macs), (geo.N1, geo.NM, geo.N2), s1.a.size, s3.c.size, geo.run_bytes)
        if before - after >= MIN_GAIN * before:
            gain3[i3] = (i1, i2, before - after)
    best = {}   # step -> (gain of the chain up to and including it, links chosen)
    zero = (0.0, ())
    for i in sorted(set(pred) | set(pred.values())):
        cands = [best.get(pred.get(i), zero)]
        if i in gain2 and gain2[i][0] == pred.get(i):
            i1, g = gain2[i]
            b = best.get(pred.get(i1), zero)
            cands.append((b[0] + g, b[1] + ((i1, i),)))
        if i in gain3:
            i1, i2, g = gain3[i]
            b = best.get(pred.get(i1), zero)
            cands.append((b[0] + g, b[1] + ((i1, i2, i),)))
        best[i] = max(cands, key=lambda c: c[0])
    chosen = {}
    ends = set(best) - set(pred.values())
    for i in sorted(ends):
        for link in best[i][1]:
            for a, b in zip(link, link[1:]):
                chosen[steps[a].node] = steps[b].node
    in_chain = {n for kv in chosen.items() for n in kv}
    if os.environ.get("CTG_NO_STEM_ONE", "0") in ("", "0"):
        for i, s1 in enumerate(steps):
            if s1.kind != P.KIND_PAIR or s1.node in in_chain or s1.a.size < min_elems or s1.a.leaf >= 0:
                continue
            if classify(i) is None:
                continue
            geo = geometry_one(size_dict, s1.a, s1.b, s1.c.inds)
            if geo is None:
                continue
            before = unfused_seconds(s1)
            after = single_seconds(s1.macs, s1.a.size, s1.c.size, geo.run_bytes, bf16x3=bf16x3, bf3_fits=geo.bf3_fits)
            if before - after >= MIN_GAIN * before:
                chosen[s1.node] = s1.node
    return chosen


def triple_shape(geo):
    """The template arguments of ``stem2_kernel`` a three-step tile needs
    (csrc/ctg_stem.hip: CTG_STEM_TRI): 16 columns in step 1 / middle / last, units per wave, column
    groups of step 1, chunks of K1, items per wave of the middle and the last step, 16-byte gathers."""
    cs1 = max(1, geo.N1 // 32)
    return (geo.N1 == 16, geo.NM == 16, geo.N2 == 16, ((1 << (geo.nr1 - 5)) * cs1) // WAVES, cs1, geo.K1 // 16,
            geo.items_m // WAVES, geo.items // WAVES, bool(geo.vec))


def _triple_instantiated(geo):
    """Three-step tiles run on static instantiations only: ask the library (a pure function of the
    shape, no device needed) -- the planner must not emit a record the kernel list does not hold."""
    from . import runtime

    if os.environ.get("CTG_STEM_TRIPLES") == "any":   # (planner tests: every tile that fits)
        return True
    lib = runtime.load()
    fn = getattr(lib, "ctg_stem_triple_instantiated", None)
    if fn is None:
        return False
    p1, pm, p2, rt1, cs1, nch, itm, it2, vec = triple_shape(geo)
    return bool(fn(int(p1), int(pm), int(p2), rt1, cs1, nch, itm, it2, int(vec)))


def build_stem_step(size_dict, A, B1, B2, c1_inds, out_inds, out_ref_factory, node=-1):
    """Lower the pair ``A, B1 -> c1_inds`` then ``C1, B2 -> out_inds`` to one
    STEM2 step (``None`` if the pair does not fit after all).  ``c1_inds`` is
    the index set of the intermediate (its order is irrelevant: it never exists
    in memory)."""
    geo = geometry(size_dict, A, B1, B2, c1_inds, out_inds)
    if geo is None:
        return None
    o2 = set(out_inds)
    keep_a2 = [ix for ix in A.inds if ix in o2] + [ix for ix in B1.inds if ix in o2 and ix not in set(A.inds)]
    keep_b2 = [ix for ix in B2.inds if ix in o2]
    natural = tuple(keep_a2) + tuple(keep_b2)
    C = out_ref_factory(tuple(out_inds), natural)

    def sa(b):
        return _stride(A, b)

    def sc(b):
        return _stride(C, b)

    K1, N1, K2, N2 = geo.K1, geo.N1, geo.K2, geo.N2
    r2 = sorted(geo.r2_members, key=sc)            # rows of step 2, digit 0 = lowest stride in C2
    n2 = sorted(geo.n2, key=sc)
    ld2 = K2 + 4

    # ---- step 1: what one wave gathers per task = 32 rows x 16 k ------------
    row_a, k_a = geo.row_a, geo.k_a
    lane, slot = np.arange(64), np.arange(8)
    if geo.vec:
        lane_a = row_a[lane & 31] + k_a[2 * (lane >> 5)]       # [64] lane part of a task's addresses
        kj_a = k_a[4 * (slot >> 1) + (slot & 1)]               # [8]  slot part (pairs adjacent in memory)
    else:
        lane_a = row_a[lane & 31] + k_a[lane >> 5]
        kj_a = k_a[2 * slot]
    rt_a = row_a[::32].copy()                                  # [2^nr1 / 32] first row of every row tile
    chunk_a = k_a[::16].copy()                                 # [K1 / 16]

    def b_off(ref, kbits, nbits):
        tk = _table(kbits, [_stride(ref, b) for b in kbits])
        tn = _table(nbits, [_stride(ref, b) for b in nbits])
        return (tk[:, None] + tn[None, :]).reshape(-1)

    b1_off = b_off(B1, geo.k1, geo.n1)                         # [k * N1 + n]
    b2_off = b_off(B2, geo.k2, n2)                             # [k2 * N2 + n2]

    # ---- the intermediate tile in LDS: element -> row2 * ld2 + k2 ------------
    pos_k2 = {b: p for p, b in enumerate(geo.k2)}
    pos_r2 = {b: p for p, b in enumerate(r2)}

    def mid_stride(b):
        return (1 << pos_k2[b]) if b in pos_k2 else (1 << pos_r2[b]) * ld2

    mid_row = _table(geo.r1, [mid_stride(b) for b in geo.r1])  # [2^nr1]
    mid_col = _table(geo.n1, [mid_stride(b) for b in geo.n1])  # [N1]
    out_row = _table(r2, [sc(b) for b in r2])                  # [rows2]
    out_col = _table(n2, [sc(b) for b in n2])                  # [N2]

    # ---- grid: one entry per tile, two-level ---------------------------------
    g_lo_bits = min(G_LO_BITS, len(geo.grid))
    glo, ghi = geo.grid[:g_lo_bits], geo.grid[g_lo_bits:]
    tabs = {
        "gA_hi": _table(ghi, [sa(b) for b in ghi]), "gA_lo": _table(glo, [sa(b) for b in glo]),
        "gC_hi": _table(ghi, [sc(b) for b in ghi]), "gC_lo": _table(glo, [sc(b) for b in glo]),
        "kj_a": kj_a, "lane_a": lane_a, "rt_a": rt_a, "chunk_a": chunk_a,
        "b1_off": b1_off, "b2_off": b2_off, "mid_row": mid_row, "mid_col": mid_col,
        "out_row": out_row, "out_col": out_col,
    }

    step = P.Step(kind=P.KIND_STEM2, kernel=P.KERNEL_MFMA, a=A, b=B1, c=C, node=node)
    step.b2 = B2
    step.stem = {
        "K1": K1, "N1": N1, "K2": K2, "N2": N2, "nr1": geo.nr1, "rows2": 1 << geo.rows2_bits,
        "ng2": geo.ng2, "n_tiles": 1 << len(geo.grid), "g_lo": 1 << g_lo_bits, "ld2": ld2,
        "lds_bytes": geo.lds, "items": geo.items, "run_bytes": geo.run_bytes, "vec": int(geo.vec), "tabs": tabs,
        "bf3_fits": bool(geo.bf3_fits),
    }
    # reporting fields: the second step's shape; work and traffic of BOTH steps as if unfused
    rows_total = A.size // K1
    step.R, step.Bt, step.K, step.N = (rows_total * N1) // K2, 1, K2, N2
    macs1 = rows_total * K1 * N1
    macs2 = step.R * K2 * N2
    c1_size = rows_total * N1
    step.macs = macs1 + macs2
    b1_elems, b2_elems = K1 * N1, K2 * N2   # (a sliced leaf's .size is that of the unsliced input)
    step.elems_rw = (A.size + b1_elems + c1_size) + (c1_size + b2_elems + step.R * N2)
    step.elems_moved = A.size + b1_elems + b2_elems + step.R * N2
    step.label = f"stem2 k{K1} n{N1} | k{K2} n{N2} rows {rows_total}"
    return step


def build_stem_one(size_dict, A, B1, out_inds, out_ref_factory, node=-1):
    """Lower ``A, B1 -> out_inds`` to a STEM2 step with the ``one`` flag: the stem kernel's first
    half alone (``None`` if the step does not fit).  Same tables as a pair's first step; the result
    goes from the accumulators straight to ``gC[tile] + out_row[tile row] + out_col[n]``."""
    geo = geometry_one(size_dict, A, B1, out_inds)
    if geo is None:
        return None
    o = set(out_inds)
    natural = tuple(ix for ix in A.inds if ix in o) + tuple(ix for ix in B1.inds if ix in o)
    C = out_ref_factory(tuple(out_inds), natural)

    def sa(b):
        return _stride(A, b)

    def sc(b):
        return _stride(C, b)

    K1, N1 = geo.K1, geo.N1
    row_a, k_a = geo.row_a, geo.k_a
    lane, slot = np.arange(64), np.arange(8)
    if geo.vec:
        lane_a = row_a[lane & 31] + k_a[2 * (lane >> 5)]
        kj_a = k_a[4 * (slot >> 1) + (slot & 1)]
    else:
        lane_a = row_a[lane & 31] + k_a[lane >> 5]
        kj_a = k_a[2 * slot]
    tk = _table(geo.k1, [_stride(B1, b) for b in geo.k1])
    tn = _table(geo.n1, [_stride(B1, b) for b in geo.n1])
    g_lo_bits = min(G_LO_BITS, len(geo.grid))
    glo, ghi = geo.grid[:g_lo_bits], geo.grid[g_lo_bits:]
    none = np.zeros(1, dtype=np.int64)
    tabs = {
        "gA_hi": _table(ghi, [sa(b) for b in ghi]), "gA_lo": _table(glo, [sa(b) for b in glo]),
        "gC_hi": _table(ghi, [sc(b) for b in ghi]), "gC_lo": _table(glo, [sc(b) for b in glo]),
        "kj_a": kj_a, "lane_a": lane_a, "rt_a": row_a[::32].copy(), "chunk_a": k_a[::16].copy(),
        "b1_off": (tk[:, None] + tn[None, :]).reshape(-1), "b2_off": none, "mid_row": none, "mid_col": none,
        "out_row": _table(geo.r1, [sc(b) for b in geo.r1]),       # [2^nr1] tile rows of C
        "out_col": _table(geo.n1, [sc(b) for b in geo.n1]),       # [N1]
    }
    step = P.Step(kind=P.KIND_STEM2, kernel=P.KERNEL_MFMA, a=A, b=B1, c=C, node=node)
    step.b2 = None
    step.stem = {
        "K1": K1, "N1": N1, "K2": 0, "N2": 0, "nr1": geo.nr1, "rows2": 0, "ng2": 0,
        "n_tiles": 1 << len(geo.grid), "g_lo": 1 << g_lo_bits, "ld2": 0, "lds_bytes": geo.lds, "items": 0,
        "run_bytes": geo.run_bytes, "vec": int(geo.vec), "one": 1, "bf3_fits": bool(geo.bf3_fits), "tabs": tabs,
    }
    rows_total = A.size // K1
    step.R, step.Bt, step.K, step.N = rows_total, 1, K1, N1
    step.macs = rows_total * K1 * N1
    step.elems_rw = A.size + K1 * N1 + rows_total * N1
    step.elems_moved = step.elems_rw
    step.label = f"stem1 k{K1} n{N1} rows {rows_total}"
    return step


def triples_enabled():
    """Three-step tiles (``geometry3`` / ``build_stem_triple``) are among ``find_pairs``' choices only
    when ``CTG_STEM_TRIPLES`` is set to something other than "" / "0".  Round 4 built and measured
    them: parity with the oracle in both arithmetics, and SLOWER than a pair
    plus a single step on every m20 tree -- in the bf16 x 3 arithmetic a pair of 16- or 32-column
    steps is already balanced between its traffic and its arithmetic, so taking the third step's
    traffic out leaves the tile bound by three steps' arithmetic (``TRIPLE_STAGE_RATE``).  Kept as an
    experiment switch for the day a stage's arithmetic gets cheaper; priced with the measured rates
    the dynamic programme rarely takes one."""
    return os.environ.get("CTG_STEM_TRIPLES", "0") not in ("", "0")


def triple_lds_bytes(K1, N1, KM, NM, K2, N2, rowsM, rows2, bf16x3=True):
    """LDS of a three-step tile: the three small operands' planes, the two intermediates in ONE
    region (the second is written over the first between two barriers), the column table
    (csrc/ctg_stem.hip: stem3_lds_bytes)."""
    mid = 8 * max(rowsM * (KM + 4), rows2 * (K2 + 4))
    if bf16x3:
        q = ((3 if N1 == 16 else 2) * N1 * ((K1 >> 4) * 48 + 8) + (3 if NM == 16 else 2) * NM * ((KM >> 3) * 24 + 8)
             + (3 if N2 == 16 else 2) * N2 * ((K2 >> 3) * 24 + 8))
        return 2 * q + mid + 8 * N2 + 64
    return b_lds_bytes(K1, N1) + b_lds_bytes(KM, NM) + b_lds_bytes(K2, N2) + mid + 8 * N2


def geometry3(size_dict, A, B1, BM, B2, c1_inds, cm_inds, out_inds):
    """Tile decomposition of THREE consecutive stem steps

        C1[r1, n1] = sum_k1 A[r1, k1] B1[k1, n1]
        CM[rM, nM] = sum_kM C1[rM, kM] BM[kM, nM]       (the middle step; rM u kM = r1 u n1)
        C2[r2, n2] = sum_k2 CM[r2, k2] B2[k2, n2]       (the last one;   r2 u k2 = rM u nM)

    as one tile: the row digits of a tile of ``A`` hold the contracted digits of BOTH later steps
    that live on ``A`` (``kMr``, ``k2a``) plus the lowest-stride free digits; the first intermediate
    is laid out ``[rM][kM]`` in LDS, the second ``[r2][k2]`` over it.  ``None`` if the three do not
    fit (shapes, tile rows, LDS, item counts that are multiples of the eight waves)."""
    o1, om, o2 = set(c1_inds), set(cm_inds), set(out_inds)
    a_bits = _bits_of(A.inds, size_dict)
    k1 = _bits_of([ix for ix in A.inds if ix not in o1], size_dict)
    n1 = _bits_of([ix for ix in B1.inds if ix in o1], size_dict)
    km = _bits_of([ix for ix in c1_inds if ix not in om], size_dict)
    nm = _bits_of([ix for ix in BM.inds if ix in om], size_dict)
    k2 = _bits_of([ix for ix in cm_inds if ix not in o2], size_dict)
    n2 = _bits_of([ix for ix in B2.inds if ix in o2], size_dict)
    if None in (a_bits, k1, n1, km, nm, k2, n2):
        return None
    K1, N1, KM, NM, K2, N2 = (1 << len(g) for g in (k1, n1, km, nm, k2, n2))
    if any(k not in K_OK for k in (K1, KM, K2)) or N1 not in N1_OK or NM not in N2_OK or N2 not in N2_OK:
        return None
    a_set, k1_set, n1_set, nm_set, km_set = set(a_bits), set(k1), set(n1), set(nm), set(km)
    kmr = [b for b in km if b not in n1_set]
    if any(b not in a_set or b in k1_set for b in kmr):
        return None
    k2a = [b for b in k2 if b in a_set]                      # contracted by the last step, still on A's rows
    if any(b in k1_set or b in km_set for b in k2a):
        return None
    if any(b not in a_set and b not in n1_set and b not in nm_set for b in k2):
        return None
    if any(b in n1_set and b in km_set for b in k2):          # (gone after the middle step)
        return None

    def sa(b):
        return _stride(A, b)

    taken = set(kmr) | set(k2a)
    free = sorted((b for b in a_bits if b not in k1_set and b not in taken), key=sa)
    cs1 = max(1, N1 // 32)
    best = None
    for units in (WAVES, 2 * WAVES):
        if units < cs1:
            continue
        nr1 = 5 + _log2(units // cs1)
        nx = nr1 - len(kmr) - len(k2a)
        if nx < 0 or nx > len(free):
            continue
        rm_bits = nr1 + len(n1) - len(km)
        r2_bits = rm_bits + len(nm) - len(k2)
        if rm_bits < 5 or r2_bits < 5:
            continue
        rows_m, rows2 = 1 << rm_bits, 1 << r2_bits
        lds = triple_lds_bytes(K1, N1, KM, NM, K2, N2, rows_m, rows2) + LDS_SLACK
        if lds > LDS_BYTES:
            continue
        ngm, ng2 = max(1, NM // 32), max(1, N2 // 32)
        items_m, items2 = (rows_m // 32) * ngm, (rows2 // 32) * ng2
        if items_m % WAVES or items2 % WAVES or items_m // WAVES > 2 or items2 // WAVES > 4:
            continue
        deep = 1 if (K1 == 16 and units == 2 * WAVES) else 0
        cand = (deep, -nr1, nr1, nx, rm_bits, r2_bits, ngm, ng2, items_m, items2, lds)
        if best is None or cand > best:
            best = cand
    if best is None:
        return None
    _, _, nr1, nx, rm_bits, r2_bits, ngm, ng2, items_m, items2, lds = best
    g = Geometry()
    g.K1, g.N1, g.KM, g.NM, g.K2, g.N2 = K1, N1, KM, NM, K2, N2
    g.nr1, g.rowsm_bits, g.rows2_bits, g.ngm, g.ng2, g.items_m, g.items, g.lds = nr1, rm_bits, r2_bits, ngm, ng2, items_m, items2, lds
    g.bf3_fits = True
    g.k1 = sorted(k1, key=sa)
    g.n1 = sorted(n1, key=lambda b: _stride(B1, b))
    x = free[:nx]
    g.r1 = sorted(kmr + k2a + x, key=sa)
    r1_set = set(g.r1)
    g.grid = sorted((b for b in a_bits if b not in k1_set and b not in r1_set), key=sa)
    g.km = [b for b in km if b in n1_set] + sorted(kmr, key=sa)      # kM index: the fresh columns first
    g.nm = sorted(nm, key=lambda b: _stride(BM, b))
    g.rm_members = [b for b in g.r1 if b not in km_set] + [b for b in g.n1 if b not in km_set]
    k2_set = set(k2)
    g.k2 = [b for b in k2 if b in nm_set] + [b for b in k2 if b in n1_set] + sorted(k2a, key=sa)
    g.r2_members = [b for b in g.rm_members if b not in k2_set] + [b for b in g.nm if b not in k2_set]
    g.n2 = n2
    g.row_a = _table(g.r1, [sa(b) for b in g.r1])
    g.k_a = _table(g.k1, [sa(b) for b in g.k1])
    g.vec = bool(g.k_a[1] == 1 and A.offset % 2 == 0)
    hbit = 2 if g.vec else 1
    if A.leaf >= 0 or 8 * int(g.row_a[31] + g.k_a[hbit]) >= 1 << 32:
        return None
    task = np.sort((g.row_a[:32, None] + g.k_a[None, :16]).reshape(-1))
    runs = np.flatnonzero(np.diff(task) != 1)
    g.run_bytes = 8 * int(runs[0] + 1 if len(runs) else len(task))
    return g


# Matrix-side rate of ONE stage of a three-step tile by its output columns, bf16 x 3 arithmetic, as
# measured on the m20 trees (profiles/r4_triples.txt): a stage splits every value of its row operand
# into limbs whatever the number of columns it is then multiplied with -- 7.3 vector instructions per
# MFMA with 16 columns, 3.7 with 32 -- so a 16-column stage runs at 81-86 TFLOP/s and a 32-column one
# at 135-146.  (The pair model needs no such table: a pair of 16-column steps is bound by its traffic
# and its arithmetic alike, and the one rate it uses reproduces measured pairs to 3-7 %.  It is the
# third step that has no traffic of its own to hide behind.)
TRIPLE_STAGE_RATE = {16: 85e12, 32: 140e12, 64: 165e12, 128: 165e12}


def triple_seconds(macs, cols, elems_a, elems_c2, run_bytes=256):
    """Modelled time of a three-step tile (bf16 x 3 arithmetic): the matrix time of its stages at
    ``TRIPLE_STAGE_RATE`` (``macs`` and ``cols`` per stage), the traffic of the big operand in and
    the LAST result out, the longer of the two plus a fifth of the shorter."""
    t_mfma = sum(8.0 * m / TRIPLE_STAGE_RATE[n] for m, n in zip(macs, cols))
    t_mem = 8.0 * elems_a / gather_rate(run_bytes) + 8.0 * elems_c2 / FUSED_STORE_RATE
    return max(t_mfma, t_mem) + FUSED_OVERLAP_LOSS * min(t_mfma, t_mem)


def build_stem_triple(size_dict, A, B1, BM, B2, c1_inds, cm_inds, out_inds, out_ref_factory, node=-1):
    """Lower three consecutive stem steps (``geometry3``) to one STEM2 step with a MIDDLE stage:
    the record of a pair (whose "second step" fields describe the LAST step) plus the middle
    step's shape, small operand and tables.  ``None`` if the three do not fit."""
    geo = geometry3(size_dict, A, B1, BM, B2, c1_inds, cm_inds, out_inds)
    if geo is None:
        return None
    o2 = set(out_inds)
    seen = set()
    natural = []
    for ref in (A, B1, BM, B2):
        for ix in ref.inds:
            if ix in o2 and ix not in seen:
                seen.add(ix)
                natural.append(ix)
    C = out_ref_factory(tuple(out_inds), tuple(natural))

    def sa(b):
        return _stride(A, b)

    def sc(b):
        return _stride(C, b)

    K1, N1, KM, NM, K2, N2 = geo.K1, geo.N1, geo.KM, geo.NM, geo.K2, geo.N2
    rm = list(geo.rm_members)                      # rows of the middle step: any fixed order (LDS only)
    r2 = sorted(geo.r2_members, key=sc)            # rows of the last step, digit 0 = lowest stride in C2
    n2 = sorted(geo.n2, key=sc)
    ldm, ld2 = KM + 4, K2 + 4
    row_a, k_a = geo.row_a, geo.k_a
    lane, slot = np.arange(64), np.arange(8)
    if geo.vec:
        lane_a = row_a[lane & 31] + k_a[2 * (lane >> 5)]
        kj_a = k_a[4 * (slot >> 1) + (slot & 1)]
    else:
        lane_a = row_a[lane & 31] + k_a[lane >> 5]
        kj_a = k_a[2 * slot]

    def b_off(ref, kbits, nbits):
        tk = _table(kbits, [_stride(ref, b) for b in kbits])
        tn = _table(nbits, [_stride(ref, b) for b in nbits])
        return (tk[:, None] + tn[None, :]).reshape(-1)

    def layout(kbits, rbits, ld):
        pos_k = {b: p for p, b in enumerate(kbits)}
        pos_r = {b: p for p, b in enumerate(rbits)}
        return lambda b: (1 << pos_k[b]) if b in pos_k else (1 << pos_r[b]) * ld

    at_m = layout(geo.km, rm, ldm)                 # first intermediate: element -> rowM * ldM + kM
    at_2 = layout(geo.k2, r2, ld2)                 # second: row2 * ld2 + k2
    g_lo_bits = min(G_LO_BITS, len(geo.grid))
    glo, ghi = geo.grid[:g_lo_bits], geo.grid[g_lo_bits:]
    tabs = {
        "gA_hi": _table(ghi, [sa(b) for b in ghi]), "gA_lo": _table(glo, [sa(b) for b in glo]),
        "gC_hi": _table(ghi, [sc(b) for b in ghi]), "gC_lo": _table(glo, [sc(b) for b in glo]),
        "kj_a": kj_a, "lane_a": lane_a, "rt_a": row_a[::32].copy(), "chunk_a": k_a[::16].copy(),
        "b1_off": b_off(B1, geo.k1, geo.n1), "b2_off": b_off(B2, geo.k2, n2),
        "mid_row": _table(geo.r1, [at_m(b) for b in geo.r1]), "mid_col": _table(geo.n1, [at_m(b) for b in geo.n1]),
        "out_row": _table(r2, [sc(b) for b in r2]), "out_col": _table(n2, [sc(b) for b in n2]),
        # the middle stage: BM[kM * NM + nM]; its result (rowM, nM) -> the second intermediate
        "bm_off": b_off(BM, geo.km, geo.nm),
        "mid2_row": _table(rm, [at_2(b) for b in rm]), "mid2_col": _table(geo.nm, [at_2(b) for b in geo.nm]),
    }
    step = P.Step(kind=P.KIND_STEM2, kernel=P.KERNEL_MFMA, a=A, b=B1, c=C, node=node)
    step.b2 = B2
    step.bm = BM
    rows_total = A.size // K1
    rows_m_total = (rows_total * N1) // KM
    step.R = (rows_m_total * NM) // K2
    step.stem = {
        "K1": K1, "N1": N1, "K2": K2, "N2": N2, "nr1": geo.nr1, "rows2": 1 << geo.rows2_bits,
        "ng2": geo.ng2, "n_tiles": 1 << len(geo.grid), "g_lo": 1 << g_lo_bits, "ld2": ld2,
        "lds_bytes": geo.lds, "items": geo.items, "run_bytes": geo.run_bytes, "vec": int(geo.vec), "tabs": tabs,
        "bf3_fits": True, "KM": KM, "NM": NM, "rowsM": 1 << geo.rowsm_bits, "ngM": geo.ngm, "ldM": ldm,
        "itemsM": geo.items_m,
    }
    step.Bt, step.K, step.N = 1, K2, N2
    macs1, macs_m, macs2 = rows_total * K1 * N1, rows_m_total * KM * NM, step.R * K2 * N2
    c1_size, cm_size = rows_total * N1, rows_m_total * NM
    step.macs = macs1 + macs_m + macs2
    step.stem["macs3"] = (macs1, macs_m, macs2)
    b1e, bme, b2e = K1 * N1, KM * NM, K2 * N2
    step.elems_rw = (A.size + b1e + c1_size) + (c1_size + bme + cm_size) + (cm_size + b2e + step.R * N2)
    step.elems_moved = A.size + b1e + bme + b2e + step.R * N2
    step.label = f"stem3 k{K1} n{N1} | k{KM} n{NM} | k{K2} n{N2} rows {rows_total}"
    return step


TAB_ORDER = ("gA_hi", "gA_lo", "gC_hi", "gC_lo", "kj_a", "lane_a", "rt_a", "chunk_a",
             "b1_off", "b2_off", "mid_row", "mid_col", "out_row", "out_col")


def serialise_stem(step, put):
    """Descriptor of a STEM2 step in the plan's table blob (``put(array) ->
    word offset``); returns the offset of its header.  Header layout
    (csrc/ctg_common.h: StemWord): magic, K1, N1, K2, N2, nr1, rows2, ng2,
    n_tiles, g_lo, ld2, lds_bytes, B2 space / offset / leaf / size, B2 producer, 16-byte gathers,
    the single-step flag (word 18: the record describes the first half alone, K2 = N2 = 0, no B2),
    then the 14 table offsets at words 20..33."""
    st = step.stem
    head = np.zeros(DESC_WORDS, dtype=np.int64)
    head[0] = DESC_MAGIC
    head[1:12] = (st["K1"], st["N1"], st["K2"], st["N2"], st["nr1"], st["rows2"], st["ng2"],
                  st["n_tiles"], st["g_lo"], st["ld2"], st["lds_bytes"])
    if step.b2 is not None:
        head[12:16] = (step.b2.space, step.b2.offset, step.b2.leaf, step.b2.size)
    else:
        head[12:16] = (0, 0, -1, 0)
    head[16] = getattr(step, "b2_prod", -1)
    head[17] = st["vec"]
    head[18] = st.get("one", 0)
    for i, name in enumerate(TAB_ORDER):
        head[20 + i] = put(st["tabs"][name])
    if st.get("KM"):
        # a middle stage (three-step tile): its shape at words 34..39, its small operand at 40..44,
        # its tables (bm_off, mid2_row, mid2_col) at 45..47
        bm = step.bm
        head[34:40] = (st["KM"], st["NM"], st["rowsM"], st["ngM"], st["ldM"], 1)
        head[40:45] = (bm.space, bm.offset, bm.leaf, bm.size, getattr(step, "bm_prod", -1))
        for i, name in enumerate(("bm_off", "mid2_row", "mid2_col")):
            head[45 + i] = put(st["tabs"][name])
    return put(head)
