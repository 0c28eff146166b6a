"""LDS-resident subtrees: the small-tree execution model (round 6).

The reference walks a contraction tree one pairwise step at a time
(``cotengra/contract.py:788-832``) and so did the executor: one launch per step
(or per wave front of independent small steps).  A step on tensors of a few
hundred elements costs such an executor what every launch costs -- a launch
plus a chain of dependent loads, 5-12 us -- whatever it computes: the
8 x 8 lattice (63 steps), the Sycamore m10 amplitude (~ 150 such steps per
slice) and the leaf levels of every other tree pay for latency, not for work.

Here the planner cuts off every *maximal subtree whose tensors all fit one
compute unit's LDS* (160 KB on gfx950).  Such a subtree -- a "component" --
is executed by ONE workgroup of ONE launch: its leaves are gathered from the
resident inputs into LDS, its steps run phase by phase (a phase = the steps
whose operands are ready) with a workgroup barrier between phases, no
intermediate ever leaves the chip, and only the subtree's root is written to
the arena.  All components of a tree are workgroups of the same launch
(``blockIdx.x``), the slices of a batch its ``blockIdx.y``.

What the planner emits (``plan.lds_runs``) is a second lowering of the member
steps -- "shadow" steps built by the same ``build_pair_step`` /
``build_single_step`` on tensors that live in LDS -- next to the ordinary
steps, which stay complete and valid: they come FIRST in the plan's step order
(so the arena is assigned for the order the launch sequence really has), and
an executor that does not take the LDS path (``strip_exponent`` keeps a scale
per step; ``CTG_NO_LDS_RUNS=1``) runs them one by one as before.

Layouts.  A tensor inside a component has no layout to honour but its
consumer's: the operand that supplies the rows of its consumer step is stored
``[contracted..., batch..., kept...]`` -- consecutive rows are consecutive LDS
words, so the lanes of a wavefront (one row each) read conflict-free at every
k --, the other operand ``[batch..., contracted..., kept...]`` (it is read as
a broadcast).  The root keeps the arena layout of the ordinary step.
"""

from __future__ import annotations

import os

import numpy as np

from .utils import prod

SPACE_LDS = 3

LDS_BYTES = 160 * 1024          # per workgroup on gfx950
LDS_TABLE_BYTES = 32 * 1024     # staged step records + offset tables (the runtime re-checks its own packing)
LDS_DATA_BYTES = LDS_BYTES - LDS_TABLE_BYTES - 1024
MAX_STEP_MACS = 1 << 19         # one compute unit does ~ 10-20 multiply-adds per clock on such steps
MAX_COMP_MACS = 3 << 18
MAX_K = 2048
MAX_COMP_STEPS = 192
MAX_EXTERNAL_ELEMS = 1 << 13    # a leaf / an invariant result larger than this is not worth a private copy

LR_MAGIC = 0x4C44535231         # "LDSR1"
LR_HEAD_WORDS = 8
LR_WORDS = 40                   # int64 words per shadow record (csrc/ctg_common.h: LdsRecWord)
KIND_LOAD, KIND_LPAIR = 0, 1


def lds_runs_enabled():
    """On unless ``CTG_LDS_RUNS`` is "0" / "" (the planner then emits no components at all)."""
    return os.environ.get("CTG_LDS_RUNS", "1") not in ("", "0")


class _FirstFit:
    """First-fit allocator over LDS element offsets (16-byte granules)."""

    def __init__(self, align):
        self.align = max(1, align)
        self.free = []
        self.top = 0
        self.peak = 0

    def _round(self, n):
        return (max(n, 1) + self.align - 1) // self.align * self.align

    def alloc(self, n):
        n = self._round(n)
        for i, (off, sz) in enumerate(self.free):
            if sz >= n:
                if sz == n:
                    self.free.pop(i)
                else:
                    self.free[i] = (off + n, sz - n)
                return off
        if self.free and self.free[-1][0] + self.free[-1][1] == self.top:
            off, _ = self.free.pop()
        else:
            off = self.top
        self.top = off + n
        self.peak = max(self.peak, self.top)
        return off

    def release(self, off, n):
        n = self._round(n)
        self.free.append((off, n))
        self.free.sort()
        merged = []
        for o, s in self.free:
            if merged and merged[-1][0] + merged[-1][1] == o:
                merged[-1] = (merged[-1][0], merged[-1][1] + s)
            else:
                merged.append((o, s))
        self.free = merged


def _table_words(R, K, N, lo_max=4096):
    """Upper bound of the table entries a shadow step stages (rows of three operands, k of two, n of two)."""
    return 3 * (min(R, lo_max) + (R + lo_max - 1) // lo_max) + 2 * K + 2 * N


def layout_in_lds(size_dict, tensor_inds, consumer):
    """Index order (slow -> fast) of a tensor stored in LDS for its consumer step:
    ``consumer = (role, other_inds, out_inds)`` with role "A" (supplies the rows) or "B"."""
    role, other, out = consumer
    inds = list(dict.fromkeys(tensor_inds))
    o_set, x_set = set(out), set(other)
    con = [ix for ix in inds if ix not in o_set]
    batch = [ix for ix in inds if ix in o_set and ix in x_set]
    keep = [ix for ix in inds if ix in o_set and ix not in x_set]
    if role == "A":
        return tuple(con + batch + keep)
    return tuple(batch + con + keep)


class Component:
    """One LDS-resident subtree: its member nodes (pair steps, in execution order), the phase of each,
    and the LDS element offset of every tensor it holds."""

    def __init__(self, root):
        self.root = root
        self.nodes = []        # member pair nodes in execution order
        self.phase = {}        # node -> phase (>= 1); loads are phase 0
        self.loads = []        # keys of tensors gathered from global memory: ("leaf", i) or ("ext", node)
        self.offset = {}       # key -> LDS element offset (key: node, ("leaf", i), ("ext", node))
        self.peak = 0
        self.macs = 0
        self.table_words = 0


def _simulate(order, kids_of, sizes, root, align, cap):
    """LDS offsets of a component's tensors.  ``order``: member nodes in execution order with their
    phase; a tensor lives from the phase that writes it (loads: phase 0) to the END of the phase that
    reads it.  Returns ``(offsets, peak)`` or None if the peak exceeds ``cap`` elements."""
    ff = _FirstFit(align)
    off = {}
    members = {n for n, _ in order}
    loads = []
    for n, _ in order:
        for c in kids_of[n]:
            if c not in members:
                loads.append(c)
    for key in loads:
        off[key] = ff.alloc(sizes[key])
    if ff.peak > cap:
        return None
    by_phase = {}
    for n, ph in order:
        by_phase.setdefault(ph, []).append(n)
    for ph in sorted(by_phase):
        for n in by_phase[ph]:
            if n != root:
                off[n] = ff.alloc(sizes[n])
        if ff.peak > cap:
            return None
        for n in by_phase[ph]:
            for c in kids_of[n]:
                ff.release(off[c], sizes[c])
    return off, ff.peak


CLASS_RANK = {"inv": 0, "group": 1, "slice": 2}


def choose_components(tree, dtype, itemsize, node_class, pairs=None):
    """Maximal subtrees of ``tree`` that run LDS-resident.

    ``node_class(node)`` -> "inv" | "group" | "slice": how often the step that makes the node's tensor
    runs (once per upload / once per slice group / every slice); for a leaf: the class of its
    preprocessing step (a leaf without one is read straight from the resident inputs and has no class of
    its own).  The members of a component all have the class of its root ("group" or "slice"); a child of
    a lower class is an external operand, gathered from the arena like a leaf is from the inputs.
    Returns a list of :class:`Component`."""
    pairs = pairs or {}
    size_dict = tree.size_dict
    align = max(1, 16 // itemsize)
    cap = LDS_DATA_BYTES // itemsize
    children = tree.children
    root_legs = tuple(ix for ix in tree.output if ix not in tree.sliced_inds)

    def legs_of(node):
        if node == tree.root:
            return root_legs
        return tuple(tree.get_legs(node))

    sizes, macs, shape = {}, {}, {}
    ok, kids_of, sub, parent_of = {}, {}, {}, {}
    topo = list(tree.traverse())
    level = {}
    for p, l, r in topo:
        parent_of[l] = parent_of[r] = p
        level[p] = 1 + max(level.get(l, 0), level.get(r, 0))
    fused = set(pairs) | set(pairs.values())

    def orders(members):
        phase = {}
        for n in members:
            phase[n] = 1 + max([phase.get(k, 0) for k in kids_of[n]])
        return phase, sorted(members, key=lambda n: (phase[n], level[n], macs[n], members.index(n)))

    for p, l, r in topo:
        ok[p] = False
        cls = node_class(p)
        if cls == "inv" or p in fused:
            continue
        l_legs, r_legs, p_legs = tuple(tree.get_legs(l)), tuple(tree.get_legs(r)), legs_of(p)
        o_set, l_set, r_set = set(p_legs), set(l_legs), set(r_legs)
        involved = list(dict.fromkeys(l_legs + r_legs))
        m = prod(size_dict[ix] for ix in involved)
        K = prod(size_dict[ix] for ix in involved if ix not in o_set)
        if m > MAX_STEP_MACS or K > MAX_K:
            continue
        keys, members, feasible = [], [], True
        for c in (l, r):
            if c not in children:           # a leaf
                pre = c in tree.preprocessing and tree.N > 1
                key = ("leaf", c) if (not pre or node_class(c) == cls) else ("ext", c)
                sizes[key] = prod(size_dict[ix] for ix in tree.get_legs(c))
                feasible = feasible and sizes[key] <= MAX_EXTERNAL_ELEMS
                keys.append(key)
            elif CLASS_RANK[node_class(c)] < CLASS_RANK[cls]:   # computed less often: read from the arena
                key = ("ext", c)
                sizes[key] = prod(size_dict[ix] for ix in legs_of(c))
                feasible = feasible and sizes[key] <= MAX_EXTERNAL_ELEMS
                keys.append(key)
            elif ok[c]:
                keys.append(c)
                members += sub[c]
            else:
                feasible = False
        if not feasible:
            continue
        sizes[p] = prod(size_dict[ix] for ix in p_legs)
        macs[p] = m
        kids_of[p] = tuple(keys)
        keep_l = prod(size_dict[ix] for ix in l_legs if ix in o_set and ix not in r_set)
        keep_r = prod(size_dict[ix] for ix in r_legs if ix in o_set and ix not in l_set)
        bt = prod(size_dict[ix] for ix in l_legs if ix in o_set and ix in r_set)
        shape[p] = (bt * max(keep_l, keep_r), K, min(keep_l, keep_r))
        members = members + [p]
        if len(members) * 3 > MAX_COMP_STEPS or sum(macs[n] for n in members) > MAX_COMP_MACS:
            continue
        twords = sum(_table_words(*shape[n]) for n in members) + sum(
            2 * sizes[k] for n in members for k in kids_of[n] if not isinstance(k, int))
        if twords * 4 + len(members) * 3 * 128 > LDS_TABLE_BYTES:
            continue
        phase, order = orders(members)
        if _simulate([(n, phase[n]) for n in order], kids_of, sizes, p, align, cap) is None:
            continue
        ok[p] = True
        sub[p] = members

    comps = []
    for p, l, r in topo:
        if not ok[p]:
            continue
        up = parent_of.get(p)
        if up is not None and ok.get(up, False):
            continue
        members = sub[p]
        comp = Component(p)
        phase, order = orders(members)
        off, peak = _simulate([(n, phase[n]) for n in order], kids_of, sizes, p, align, cap)
        comp.cls = node_class(p)
        comp.nodes = order
        comp.phase = phase
        comp.offset = off
        comp.peak = peak
        comp.macs = sum(macs[n] for n in members)
        comps.append(comp)
    return comps


def serialise_run(run, put):
    """Descriptor of one component in the plan's table blob; returns the word offset of its header.

    Header (LR_HEAD_WORDS): magic, number of shadow records, LDS data elements needed, number of
    phases, component id.  Then one record of LR_WORDS per shadow step (csrc/ctg_common.h: LdsRecWord):
      0 kind (0 load: global -> LDS gather with optional sum, 1 pair)   1 phase
      2 main step the record belongs to   3 which operand of that step is the GLOBAL side (0 A, 1 B, 2 C, -1 none)
      4/5 A in LDS? / LDS element offset   6/7 B   8/9 C
      10 R  11 K  12 N  13 row_lo  14 row_hi_len  15 k_lo  16 k_hi_len
      17..22 rowA_hi rowA_lo rowB_hi rowB_lo rowC_hi rowC_lo   23..26 kA_hi kA_lo kB_hi kB_lo   27 nB  28 nC
      29..31 addressable elements of A, B, C   32 macs
    """
    recs = np.zeros((len(run["steps"]), LR_WORDS), dtype=np.int64)
    zero = put(np.zeros(1, dtype=np.int64))
    for i, sh in enumerate(run["steps"]):
        s = sh["step"]
        r = recs[i]
        r[0], r[1], r[2], r[3] = sh["kind"], sh["phase"], sh["main"], sh["global_operand"]
        for base, t in ((4, s.a), (6, s.b), (8, s.c)):
            if t is not None and t.space == SPACE_LDS:
                r[base], r[base + 1] = 1, t.offset
            else:
                r[base], r[base + 1] = 0, 0
        r[10], r[11], r[12] = s.R, s.K, s.N
        r[13] = s.row_lo
        hi_len = 1
        for j, key in enumerate("ABC"):
            if key in s.rows:
                hi, lo = s.rows[key]
                hi_len = len(hi)
                r[17 + 2 * j], r[18 + 2 * j] = put(hi), put(lo)
            else:
                r[17 + 2 * j] = r[18 + 2 * j] = -1
        r[14] = hi_len
        r[15], r[16] = s.k_lo, 1
        r[23] = r[24] = r[25] = r[26] = zero
        for key, w_hi, w_lo in (("A", 23, 24), ("B", 25, 26)):
            if key in s.k_tabs:
                hi, lo = s.k_tabs[key]
                r[w_hi], r[w_lo] = put(hi), put(lo)
                r[16] = len(hi)
        r[27] = put(s.n_tabs["B"]) if "B" in s.n_tabs else zero
        r[28] = put(s.n_tabs["C"]) if "C" in s.n_tabs else zero
        r[29] = s.a.size if s.a is not None else 0
        r[30] = s.b.size if s.b is not None else 0
        r[31] = s.c.size if s.c is not None else 0
        r[32] = s.macs
    head = np.zeros(LR_HEAD_WORDS, dtype=np.int64)
    head[0], head[1], head[2], head[3], head[4] = LR_MAGIC, len(run["steps"]), run["lds_elems"], run["n_phases"], run["id"]
    return put(np.concatenate([head, recs.reshape(-1)]))


def build_shadows(plan, tree, dtype, comps, step_of_node, operand_node):
    """Second lowering of the member steps of every component, on LDS-resident tensors
    (``plan.lds_runs``), and the member flags of the ordinary steps (``Step.lds_comp``).

    ``step_of_node``: tree node -> its ordinary PAIR step; ``operand_node``: id(TensorRef) -> the tree
    node (or leaf) whose tensor the reference is."""
    from .plan import (KERNEL_VALU, KIND_PAIR, KIND_SINGLE, TensorRef, _row_major_strides, build_pair_step,
                       build_single_step)

    size_dict = tree.size_dict
    index_of = {id(s): i for i, s in enumerate(plan.steps)}
    consumer_of = {}
    for s in plan.steps:
        if s.kind == KIND_PAIR:
            consumer_of[id(s.a)] = (s, "A")
            consumer_of[id(s.b)] = (s, "B")
    single_of_leaf = {s.node: s for s in plan.steps if s.kind == KIND_SINGLE and s.node >= 0 and s.node < tree.N}

    def step_class(st):
        return "inv" if st.invariant else ("group" if st.group else "slice")

    def lds_ref(inds, offset):
        shape = [size_dict[ix] for ix in inds]
        return TensorRef(SPACE_LDS, int(offset), -1, tuple(inds), _row_major_strides(shape), prod(shape))

    def wanted_layout(main_ref):
        """Index order of the LDS copy of ``main_ref`` (an operand of a member step)."""
        s, role = consumer_of[id(main_ref)]
        other = s.b if role == "A" else s.a
        return layout_in_lds(size_dict, main_ref.inds, (role, other.inds, s.c.inds))

    for cid, comp in enumerate(comps):
        members = set(comp.nodes)
        shadows = []
        have = {}   # key -> LDS TensorRef
        main_members = []
        # phase 0: everything the component reads from global memory
        for n in comp.nodes:
            ms = step_of_node[n]
            for role, ref in (("A", ms.a), ("B", ms.b)):
                child = operand_node[id(ref)]
                if child in members:
                    continue
                single = single_of_leaf.get(child) if child < tree.N else None
                member_single = single is not None and step_class(single) == comp.cls
                key = ("leaf", child) if (child < tree.N and (single is None or member_single)) else ("ext", child)
                order = wanted_layout(ref)
                off = comp.offset[key]
                if member_single:
                    # the leaf's own preprocessing (diagonal / trace / sum), gathered straight into LDS
                    sh = build_single_step(size_dict, single.a, order, lambda inds, nat: lds_ref(inds, off), node=child)
                    main, gop = index_of[id(single)], 0
                    main_members.append(main)
                else:
                    sh = build_single_step(size_dict, ref, order, lambda inds, nat: lds_ref(inds, off), node=child)
                    main, gop = index_of[id(ms)], (0 if role == "A" else 1)
                have[id(ref)] = sh.c
                shadows.append({"kind": KIND_LOAD, "phase": 0, "main": main, "global_operand": gop, "step": sh})
        for n in comp.nodes:
            ms = step_of_node[n]
            is_root = n == comp.root

            def factory(inds, natural, _ms=ms, _root=is_root, _n=n):
                if _root:
                    return _ms.c
                return lds_ref(wanted_layout(_ms.c), comp.offset[_n])

            sh = build_pair_step(dtype, size_dict, have[id(ms.a)], have[id(ms.b)], ms.c.inds, factory, node=n,
                                 force_kernel=KERNEL_VALU)
            have[id(ms.c)] = sh.c
            main = index_of[id(ms)]
            main_members.append(main)
            shadows.append({"kind": KIND_LPAIR, "phase": comp.phase[n], "main": main,
                            "global_operand": 2 if is_root else -1, "step": sh})
        for m in main_members:
            plan.steps[m].lds_comp = cid
        plan.lds_runs.append({
            "id": cid,
            "steps": shadows,
            "lds_elems": int(comp.peak),
            "n_phases": 1 + max(comp.phase.values()),
            "members": sorted(main_members),
            "root_main": index_of[id(step_of_node[comp.root])],
            "cls": comp.cls,
            "macs": comp.macs,
        })
