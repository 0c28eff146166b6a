"""Host-side contraction tree: the *metadata* half of cotengra's
``ContractionTree`` that the MI355X executor needs, written from scratch.

A tree here is an immutable-by-convention binary merge schedule over ``N``
input tensors, labelled with SSA integers exactly like the reference's default
``NodeOpsSSA`` (reference ``cotengra/nodeops.py:256-332``: leaves ``0..N-1``,
root ``N``, intermediates ``N+1, N+2, ...`` in creation order).  Everything
else (legs, index order, sizes, costs, tensordot axes, einsum equations, slice
keys) is *derived* and recomputed lazily whenever the set of sliced indices
changes, instead of being patched incrementally as the reference does
(``cotengra/core.py:1966-2042``) -- the results are identical, the code is a
few hundred lines and has no hidden state.

Only what the execution path reads is provided (SURVEY.md section 8a rows a1-a7,
a13-a20): searching for trees, reconfiguring them and choosing which indices
to slice stay in the reference's host-side optimizers, whose *output*
(``path`` + ``sliced_inds``) is this class's input.

Method names, argument meaning and error behaviour follow the reference so
that code written against ``cotengra.ContractionTree`` keeps working:

=====================  =======================================================
this file              reference (cotengra v0.8.2)
=====================  =======================================================
``SliceInfo``          ``core.py:99-111``
``get_slice_strides``  ``core.py:114-122``
``from_path``          ``core.py:537-636`` (n-ary steps expanded greedily)
``get_legs`` ...       ``core.py:861-1095``
``traverse``           ``core.py:1781-1864``
``remove_ind``         ``core.py:1966-2042``
``restore_ind``        ``core.py:2046-2089``
``get_path``           ``core.py:3188-3254``
``total_flops`` ...    ``core.py:1196-1364``
``slice_key`` ...      ``core.py:3775-3941``
``contract`` ...       ``core.py:3638-3773, 3943-4164``
=====================  =======================================================
"""

from __future__ import annotations

import functools
import itertools
import math
import warnings
from dataclasses import dataclass
from typing import Optional

from .utils import (
    get_symbol,
    inputs_output_to_eq,
    prod,
    unique,
)


@dataclass(order=True, frozen=True)
class SliceInfo:
    """One removed index.  Field order defines the sort order, which is what
    puts output ('outer') sliced indices first (reference core.py:99-111,
    1986-1991)."""

    inner: bool
    ind: str
    size: int
    project: Optional[int]

    @property
    def sliced_range(self):
        if self.project is None:
            return range(self.size)
        return [self.project]


def get_slice_strides(sliced_inds):
    """Mixed-radix strides of the (ordered) sliced indices: slice id
    ``i = sum_j value_j * stride_j`` (reference core.py:114-122)."""
    sizes = [si.size for si in sliced_inds.values()]
    strides = [1] * len(sizes)
    for j in range(len(sizes) - 2, -1, -1):
        strides[j] = strides[j + 1] * sizes[j + 1]
    return strides


class ContractionTree:
    """Binary contraction tree over ``inputs -> output`` with optional sliced
    indices.

    Parameters
    ----------
    inputs : sequence of sequence of str
        Index labels of every input tensor (one hashable label per axis).
    output : sequence of str
        Index labels of the output tensor.
    size_dict : dict[str, int]
        Extent of every index.
    """

    # ------------------------------------------------------------------ #
    # construction
    # ------------------------------------------------------------------ #

    def __init__(self, inputs, output, size_dict):
        self.inputs = tuple(tuple(term) for term in inputs)
        self.output = tuple(output)
        self.size_dict = {k: int(v) for k, v in size_dict.items()}
        self.N = len(self.inputs)
        if self.N == 0:
            raise ValueError("A contraction needs at least one input.")

        # how often each index appears over inputs + output: an index survives
        # on an intermediate as long as it has not been seen this many times
        # (reference core.py:246-258) -- this is what makes hyper indices work
        self.appearances = {}
        for term in self.inputs:
            for ix in term:
                self.appearances[ix] = self.appearances.get(ix, 0) + 1
        for ix in self.output:
            self.appearances[ix] = self.appearances.get(ix, 0) + 1
        for ix in self.appearances:
            if ix not in self.size_dict:
                raise KeyError(f"Index {ix!r} has no size in size_dict.")

        self.root = self.N
        self._next_ssa = self.N + 1
        # parent -> (left, right)
        self.children = {}
        # node -> number of leaves below it
        self._extent = {i: 1 for i in range(self.N)}
        self._extent[self.root] = self.N
        if self.N == 1:
            self.children[self.root] = (0,)

        self.multiplicity = 1
        self.sliced_inds = {}
        self.sliced_inputs = frozenset()

        # derived, cleared by _invalidate()
        self._info = {}
        self.preprocessing = {}
        self.contraction_cores = {}

    # -- SSA bookkeeping ------------------------------------------------- #

    def gen_leaves(self):
        return iter(range(self.N))

    def input_to_node(self, i):
        return i

    def node_to_input(self, node):
        return node

    def is_leaf(self, node):
        return 0 <= node < self.N

    def is_root(self, node):
        return node == self.root

    def get_extent(self, node):
        return self._extent[node]

    def _merge(self, x, y, parent=None):
        """Create the parent of ``x`` and ``y``.  The child spanning more
        leaves goes left; equal spans are ordered by the SSA tie-breaker
        ``-node`` (reference core.py:1633-1647, nodeops.py:288-289)."""
        nx, ny = self._extent[x], self._extent[y]
        if parent is None:
            if nx + ny == self.N:
                parent = self.root
            else:
                parent = self._next_ssa
                self._next_ssa += 1
        if nx == ny:
            kx, ky = -x, -y
        else:
            kx, ky = nx, ny
        self.children[parent] = (x, y) if kx > ky else (y, x)
        self._extent[parent] = nx + ny
        return parent

    @classmethod
    def from_path(
        cls,
        inputs,
        output,
        size_dict,
        *,
        path=None,
        ssa_path=None,
        autocomplete="auto",
    ):
        """Build a complete tree from a pairwise contraction path, either
        with recycled linear ids (``path``) or single-static-assignment ids
        (``ssa_path``) -- reference core.py:537-636.  A step that merges three
        or more tensors is expanded into pairwise merges by a small greedy rule
        (smallest intermediate first); the reference runs a sub-optimizer there
        (core.py:1690), so for such steps the pairwise order -- not the result --
        may differ from the reference's.
        """
        if (path is None) == (ssa_path is None):
            raise ValueError(
                "Exactly one of ``path`` or ``ssa_path`` must be supplied."
            )
        tree = cls(inputs, output, size_dict)
        if tree.N == 1:
            return tree

        def merged_size(x, y):
            lx, ly = tree.get_legs(x), tree.get_legs(y)
            size = 1
            for ix in {**lx, **ly}:
                if lx.get(ix, 0) + ly.get(ix, 0) < tree.appearances[ix]:
                    size *= tree.size_dict[ix]
            return size

        def merge(group):
            group = list(group)
            if len(group) == 1:
                return group[0]
            while len(group) > 2:
                _, i, j = min(
                    (merged_size(group[i], group[j]), i, j)
                    for i in range(len(group)) for j in range(i + 1, len(group))
                )
                y, x = group.pop(j), group.pop(i)
                group.append(tree._merge(x, y))
            return tree._merge(*group)

        if ssa_path is not None:
            nodes = dict(enumerate(tree.gen_leaves()))
            ssa = len(nodes)
            for p in ssa_path:
                nodes[ssa] = merge([nodes.pop(int(i)) for i in p])
                ssa += 1
            nodes = list(nodes.values())
        else:
            nodes = list(tree.gen_leaves())
            for p in path:
                group = [nodes.pop(int(i)) for i in sorted(p, reverse=True)]
                nodes.append(merge(group))

        if len(nodes) > 1:
            if not autocomplete:
                raise ValueError("Path is not complete.")
            if autocomplete == "auto":
                warnings.warn(
                    "Path was not complete - contracting all remaining "
                    "tensors left to right."
                )
            while len(nodes) > 1:
                y = nodes.pop()
                x = nodes.pop()
                nodes.append(tree._merge(x, y))

        if not tree.is_complete():
            raise ValueError("Path did not produce a complete tree.")
        return tree

    @classmethod
    def from_eq(cls, eq, size_dict, **kwargs):
        lhs, output = eq.split("->")
        return cls(lhs.split(","), output, size_dict, **kwargs)

    def is_complete(self):
        if self.N == 1:
            return True
        stack = [self.root]
        seen = 0
        while stack:
            node = stack.pop()
            if self.is_leaf(node):
                seen += 1
                continue
            if node not in self.children:
                return False
            stack.extend(self.children[node])
        return seen == self.N

    def copy(self):
        new = object.__new__(self.__class__)
        new.inputs = self.inputs
        new.output = self.output
        new.size_dict = self.size_dict
        new.N = self.N
        new.appearances = self.appearances
        new.root = self.root
        new._next_ssa = self._next_ssa
        new.children = dict(self.children)
        new._extent = dict(self._extent)
        new.multiplicity = self.multiplicity
        new.sliced_inds = dict(self.sliced_inds)
        new.sliced_inputs = self.sliced_inputs
        new._info = {}
        new.preprocessing = {}
        new.contraction_cores = {}
        return new

    # ------------------------------------------------------------------ #
    # equations and shapes
    # ------------------------------------------------------------------ #

    def get_eq(self):
        return inputs_output_to_eq(self.inputs, self.output)

    def get_shapes(self):
        return tuple(
            tuple(self.size_dict[ix] for ix in term) for term in self.inputs
        )

    def get_inputs_sliced(self):
        return tuple(
            tuple(ix for ix in term if ix not in self.sliced_inds)
            for term in self.inputs
        )

    def get_output_sliced(self):
        return tuple(ix for ix in self.output if ix not in self.sliced_inds)

    def get_eq_sliced(self):
        return inputs_output_to_eq(
            self.get_inputs_sliced(), self.get_output_sliced()
        )

    def get_shapes_sliced(self):
        return tuple(
            tuple(self.size_dict[ix] for ix in term)
            for term in self.get_inputs_sliced()
        )

    # ------------------------------------------------------------------ #
    # per-node index algebra (all derived, cached in self._info)
    # ------------------------------------------------------------------ #

    def _invalidate(self):
        self._info = {}
        self.preprocessing = {}
        self.contraction_cores = {}

    def _cached(self, node, key, compute):
        d = self._info.setdefault(node, {})
        try:
            return d[key]
        except KeyError:
            value = d[key] = compute(node)
            return value

    def compute_leaf_legs(self, i):
        """Effective indices of input ``i`` after slicing and after the
        single-term simplification that removes repeated indices (diagonals /
        traces) and indices that appear nowhere else (immediate sums).  When
        such a simplification is needed its canonical einsum equation is
        recorded in ``self.preprocessing[i]`` (reference core.py:861-904).
        """
        term = tuple(ix for ix in self.inputs[i] if ix not in self.sliced_inds)
        legs = {}
        for ix in term:
            legs[ix] = legs.get(ix, 0) + 1

        simplifiable = len(term) != len(legs) or any(
            count == self.appearances[ix] for ix, count in legs.items()
        )
        if simplifiable:
            legs = {
                ix: count
                for ix, count in legs.items()
                if count != self.appearances[ix]
            }
            self.preprocessing[i] = inputs_output_to_eq(
                (term,), tuple(legs), canonicalize=True
            )
        return legs

    def has_preprocessing(self):
        for leaf in self.gen_leaves():
            self.get_legs(leaf)
        return bool(self.preprocessing)

    def has_hyper_indices(self):
        return any(c != 2 for c in self.appearances.values())

    def get_legs(self, node):
        """Ordered mapping index -> number of times seen so far below
        ``node`` (reference core.py:969-999)."""

        def compute(node):
            if node == self.root:
                return {
                    ix: 0 for ix in self.output if ix not in self.sliced_inds
                }
            if self.is_leaf(node):
                return self.compute_leaf_legs(node)
            return {
                ix: count
                for ix, count in self.get_involved(node).items()
                if count < self.appearances[ix]
            }

        return self._cached(node, "legs", compute)

    def get_involved(self, node):
        """All indices taking part in the pairwise contraction that forms
        ``node`` (reference core.py:1001-1007)."""

        def compute(node):
            if self.is_leaf(node):
                return {}
            involved = {}
            for child in self.children[node]:
                for ix, count in self.get_legs(child).items():
                    involved[ix] = involved.get(ix, 0) + count
            return involved

        return self._cached(node, "involved", compute)

    def get_size(self, node):
        return self._cached(
            node,
            "size",
            lambda n: prod(self.size_dict[ix] for ix in self.get_legs(n)),
        )

    def get_flops(self, node):
        """Scalar multiply-adds of the pairwise step forming ``node``
        (reference core.py:1014-1022)."""

        def compute(node):
            if self.is_leaf(node):
                return 0
            return prod(self.size_dict[ix] for ix in self.get_involved(node))

        return self._cached(node, "flops", compute)

    def get_can_dot(self, node):
        """True iff the step is a plain tensordot: the parent's indices are
        exactly the symmetric difference of the children's (reference
        core.py:1024-1032)."""

        def compute(node):
            l, r = self.children[node]
            sp, sl, sr = (set(self.get_legs(n)) for n in (node, l, r))
            return sp == sl.symmetric_difference(sr)

        return self._cached(node, "can_dot", compute)

    def get_inds_tuple(self, node):
        """Axis order of the tensor at ``node`` as a tuple of labels.  Leaves
        and root follow their legs; an intermediate keeps its left child's
        surviving indices followed by the right child's new ones (reference
        core.py:1034-1051)."""

        def compute(node):
            legs = self.get_legs(node)
            if self.is_leaf(node) or node == self.root:
                return tuple(legs)
            l, r = self.children[node]
            chain = itertools.chain(
                self.get_inds_tuple(l), self.get_inds_tuple(r)
            )
            return tuple(unique(ix for ix in chain if ix in legs))

        return self._cached(node, "inds", compute)

    def get_inds(self, node):
        return "".join(self.get_inds_tuple(node))

    def get_tensordot_axes(self, node):
        """``axes`` for a tensordot forming ``node``: contracted pairs in
        order of appearance on the left child (reference core.py:1053-1066).
        """

        def compute(node):
            l, r = self.children[node]
            l_inds, r_inds = self.get_inds_tuple(l), self.get_inds_tuple(r)
            r_pos = {ix: j for j, ix in enumerate(r_inds)}
            l_axes, r_axes = [], []
            for i, ix in enumerate(l_inds):
                j = r_pos.get(ix)
                if j is not None:
                    l_axes.append(i)
                    r_axes.append(j)
            return tuple(l_axes), tuple(r_axes)

        return self._cached(node, "tensordot_axes", compute)

    def get_tensordot_perm(self, node):
        """Permutation taking tensordot's ``[free-left..., free-right...]``
        output to ``get_inds(node)``, or None (reference core.py:1068-1080).
        """

        def compute(node):
            l, r = self.children[node]
            lr = self.get_inds_tuple(l) + self.get_inds_tuple(r)
            p_inds = self.get_inds_tuple(node)
            td_inds = tuple(sorted(p_inds, key=lr.index))
            if td_inds == p_inds:
                return None
            return tuple(td_inds.index(ix) for ix in p_inds)

        return self._cached(node, "tensordot_perm", compute)

    def get_einsum_eq(self, node):
        """Pairwise einsum equation for ``node`` with labels remapped into
        ``[a-zA-Z...]`` in order of first appearance (reference
        core.py:1082-1095)."""

        def compute(node):
            l, r = self.children[node]
            li, ri, pi = (self.get_inds_tuple(n) for n in (l, r, node))
            table = {
                ix: get_symbol(i)
                for i, ix in enumerate(unique(itertools.chain(li, ri)))
            }
            fmt = lambda inds: "".join(table[ix] for ix in inds)  # noqa: E731
            return f"{fmt(li)},{fmt(ri)}->{fmt(pi)}"

        return self._cached(node, "einsum_eq", compute)

    # ------------------------------------------------------------------ #
    # traversal and paths
    # ------------------------------------------------------------------ #

    def get_default_order(self):
        return "dfs"

    def _traverse_dfs(self):
        """Post-order walk: a node is emitted after its whole left subtree and
        then its whole right subtree -- the sequence the reference's
        depth-first traversal produces (core.py:1781-1799), which the parity
        of the linear IR depends on.  Each frame on the explicit stack carries
        how many of the node's children have been walked already."""
        children = self.children
        frames = [(self.root, 0)]
        while frames:
            node, walked = frames.pop()
            kids = children.get(node)
            if kids is None:  # a leaf: nothing to contract
                continue
            if walked == 2:
                yield (node, *kids)
                continue
            frames.append((node, walked + 1))
            frames.append((kids[walked], 0))

    def _traverse_ordered(self, order):
        """The reference's ordered traversal, sequence for sequence
        (core.py:1801-1832) -- ``get_path(order=f)``, ``peak_size(order=f)``
        and ``print_contractions`` are index outputs and must agree with it.

        The schedule is a line-up that starts as just the root and is swept
        from the front again and again.  A contraction met for the first time
        is *opened*: each of its children that is itself a contraction is
        slotted in somewhere in front of it -- at the upper-bound position of
        its score in the part of the line-up ahead of the parent, found by
        binary search.  That part is in general NOT sorted (a parent keeps its
        place when its own children go in front of it), so the outcome is
        defined by the bisection itself and the stdlib's ``bisect_right`` is
        used on the live list with ``hi = position of the parent``.  Children
        land ahead of the sweep position and wait for the next sweep; once
        every contraction has been opened the line-up is the order.

        The planner's *own* level order (plan.compile_tree) does not use
        this."""
        from bisect import bisect_right

        children = self.children
        if order == "surface_order":
            order = self.surface_order

        lineup = [self.root]
        keys = [order(self.root)]
        opened = set()
        while len(opened) < len(children):
            at = 0
            while at < len(lineup):
                node = lineup[at]
                if node not in opened:
                    opened.add(node)
                    for kid in children[node]:
                        if kid not in children:  # a leaf: nothing to schedule
                            continue
                        key = order(kid)
                        slot = bisect_right(keys, key, 0, at)
                        keys.insert(slot, key)
                        lineup.insert(slot, kid)
                        at += 1  # the parent has been pushed one place back
                at += 1
        for node in lineup:
            yield (node, *children[node])

    def surface_order(self, node):
        """Score used by ``order="surface_order"``.  The reference's default
        (core.py:3261: extent, then hypergraph centrality) belongs to its
        compressed-contraction machinery, which is out of scope;
        what is supported is the explicit form, an order installed from a
        path with ``set_surface_order_from_path``."""
        raise NotImplementedError(
            "order='surface_order' needs set_surface_order_from_path(ssa_path) "
            "first: centrality-based surface orders are part of the "
            "reference's compressed contraction, which this package does not "
            "rebuild."
        )

    def set_surface_order_from_path(self, ssa_path):
        """Score every contraction by its position in ``ssa_path`` (those the
        path does not produce: +inf) -- reference core.py:3264-3283."""
        by_pair = {frozenset(kids): p for p, kids in self.children.items()}
        node_of = dict(enumerate(self.gen_leaves()))
        rank = {}
        for step, ids in enumerate(ssa_path):
            parent = by_pair[frozenset(node_of[i] for i in ids)]
            rank[parent] = step
            node_of[self.N + step] = parent
        self.surface_order = lambda node: rank.get(node, float("inf"))

    def get_path_surface(self):
        return self.get_path(order=self.surface_order)

    def get_ssa_path_surface(self):
        return self.get_ssa_path(order=self.surface_order)

    def traverse(self, order=None):
        """Generate ``(parent, left, right)`` merges bottom-up."""
        if self.N == 1:
            return
        if order is None:
            order = self.get_default_order()
        if order == "dfs":
            yield from self._traverse_dfs()
        elif callable(order) or order == "surface_order":
            yield from self._traverse_ordered(order)
        else:
            raise ValueError(f"Unknown traversal order {order!r}.")

    def get_path(self, order=None):
        """Linear (recycled-id) path (reference core.py:3188-3226)."""
        from bisect import bisect_left

        ssa = self.N
        live = list(range(ssa))
        where = {leaf: leaf for leaf in self.gen_leaves()}
        path = []
        for parent, l, r in self.traverse(order=order):
            i, j = sorted(
                (bisect_left(live, where[l]), bisect_left(live, where[r]))
            )
            live.pop(j)
            live.pop(i)
            path.append((i, j))
            live.append(ssa)
            where[parent] = ssa
            ssa += 1
        return tuple(path)

    def get_ssa_path(self, order=None):
        """SSA path (reference core.py:3236-3258)."""
        pos = {leaf: leaf for leaf in self.gen_leaves()}
        ssa_path = []
        for parent, l, r in self.traverse(order=order):
            ssa_path.append(tuple(sorted((pos[l], pos[r]))))
            pos[parent] = len(ssa_path) + self.N - 1
        return tuple(ssa_path)

    # ------------------------------------------------------------------ #
    # cost model (per the reference's accounting)
    # ------------------------------------------------------------------ #

    def _flops_one_slice(self):
        return sum(self.get_flops(p) for p, _, _ in self.traverse())

    def _write_one_slice(self):
        return sum(self.get_size(p) for p, _, _ in self.traverse())

    def total_flops(self, dtype=None, log=None):
        """Scalar operations over ALL slices; x2 for float, x4 for complex
        dtypes -- the reference's convention (core.py:1196-1227), note a
        complex multiply-add is really 8 real flops."""
        C = self.multiplicity * self._flops_one_slice()
        if dtype is None:
            pass
        elif "float" in dtype:
            C *= 2
        elif "complex" in dtype:
            C *= 4
        else:
            raise ValueError(f"Unknown dtype {dtype}")
        if log is not None:
            C = math.log(max(C, 1), log)
        return C

    def contraction_cost(self, log=None):
        return self.total_flops(dtype=None, log=log)

    def total_write(self):
        return self.multiplicity * self._write_one_slice()

    def max_size(self, log=None):
        if self.N == 1:
            size = self.get_size(self.root)
        else:
            size = max(self.get_size(p) for p, _, _ in self.traverse())
        if log is not None:
            size = math.log(size, log)
        return size

    def peak_size(self, order=None, log=None):
        """Largest number of simultaneously live elements when every step
        holds both operands and its result (reference core.py:1299-1316)."""
        live = sum(self.get_size(leaf) for leaf in self.gen_leaves())
        peak = live
        for p, l, r in self.traverse(order=order):
            live += self.get_size(p)
            peak = max(peak, live)
            live -= self.get_size(l) + self.get_size(r)
        if log is not None:
            peak = math.log(peak, log)
        return peak

    def contract_stats(self, force=False):
        return {
            "flops": max(self.total_flops(), 1),
            "write": max(self.total_write(), 1),
            "size": max(self.max_size(), 1),
        }

    def arithmetic_intensity(self):
        return self.total_flops(dtype=None) / self.total_write()

    # ------------------------------------------------------------------ #
    # slicing state
    # ------------------------------------------------------------------ #

    @property
    def nslices(self):
        return self.multiplicity

    @property
    def nchunks(self):
        return prod(
            si.size for si in self.sliced_inds.values() if not si.inner
        )

    def remove_ind(self, ind, project=None, inplace=False):
        """Slice (or, with ``project=j``, fix to value ``j``) index ``ind``
        (reference core.py:1966-2042)."""
        tree = self if inplace else self.copy()
        if ind in tree.sliced_inds:
            raise ValueError(f"Index {ind} already sliced.")
        d = tree.size_dict[ind]
        inner = ind not in tree.output
        if project is None:
            si = SliceInfo(inner, ind, d, None)
            tree.multiplicity *= d
        else:
            si = SliceInfo(inner, ind, 1, int(project))
        tree.sliced_inds = {
            s.ind: s for s in sorted((*tree.sliced_inds.values(), si))
        }
        tree.sliced_inputs = tree.sliced_inputs | frozenset(
            i for i, term in enumerate(tree.inputs) if ind in term
        )
        tree._invalidate()
        return tree

    remove_ind_ = functools.partialmethod(remove_ind, inplace=True)

    def apply_slicing_(self, sliced):
        """Re-apply a slicing taken from another tree or from a record, keeping
        projections: ``sliced`` is a mapping ``ind -> SliceInfo``-like (anything
        with ``.project``, e.g. ``other.sliced_inds`` of this package or of the
        reference), or an iterable of index names / ``(name, project)`` pairs
        (the form :meth:`slicing_record` writes)."""
        if hasattr(sliced, "items"):
            items = [(ix, getattr(si, "project", None)) for ix, si in sliced.items()]
        else:
            items = [
                (e, None) if isinstance(e, str) or not isinstance(e, (tuple, list)) else (e[0], e[1])
                for e in sliced
            ]
        for ix, project in items:
            self.remove_ind_(ix, project=project)
        return self

    def gathered_shape(self):
        """Shape of what :meth:`contract` returns: the output indices at full
        extent, except that an index projected onto one value keeps a size-1
        axis (reference ``gather_slices``, core.py:3866-3876)."""
        return tuple(
            1 if (ix in self.sliced_inds and self.sliced_inds[ix].project is not None) else self.size_dict[ix]
            for ix in self.output
        )

    def slicing_record(self):
        """JSON-able form of the slicing: index names, ``[name, j]`` for an
        index projected onto value ``j``."""
        return [
            ix if si.project is None else [ix, si.project] for ix, si in self.sliced_inds.items()
        ]

    def slice(self, target_size=None, target_slices=None, allow_outer=True, inplace=False, **_ignored):
        """Remove indices until the largest intermediate has at most
        ``target_size`` elements (reference ``ContractionTree.slice``,
        core.py:2632-2719).  The indices are chosen by the native greedy finder
        (``cotengra_amd.pathfind``), not by the reference's randomised
        ``SliceFinder``; ``target_slices`` is honoured by halving the target
        until enough slices exist."""
        from .pathfind import find_sliced_inds

        if target_size is None and target_slices is None:
            raise ValueError("Need one of ``target_size`` or ``target_slices``.")
        tree = self if inplace else self.copy()
        size = tree.max_size() if target_size is None else target_size
        while True:
            for ix in find_sliced_inds(tree, size, allow_outer=allow_outer):
                tree.remove_ind_(ix)
            if target_slices is None or tree.nslices >= target_slices or size <= 1:
                return tree
            size = max(size // 2, 1)

    slice_ = functools.partialmethod(slice, inplace=True)

    def subtree_reconfigure(self, subtree_size=8, subtree_search="bfs", weight_what="flops",
                            weight_pwr=2, select="max", maxiter="auto", seed=None, minimize="flops",
                            optimize=None, inplace=False, progbar=False):
        """Reference ``ContractionTree.subtree_reconfigure`` (core.py:2316-2449) on
        the native dynamic-programming routine (``cotengra_amd.pathfind``): subtrees
        are grown breadth-first and visited most expensive first -- the reference's
        defaults; its sampling options are accepted and ignored.  ``minimize`` also
        takes ``"time"`` / a ``pathfind.MachineModel``."""
        from .pathfind import subtree_reconfigure

        return subtree_reconfigure(self, subtree_size=subtree_size, maxiter=maxiter, minimize=minimize,
                                   inplace=inplace)

    subtree_reconfigure_ = functools.partialmethod(subtree_reconfigure, inplace=True)

    def slice_and_reconfigure(self, target_size, step_size=2, temperature=0.01, minimize="flops",
                              allow_outer=True, max_repeats=16, reslice=False, reconf_opts=None,
                              progbar=False, inplace=False):
        """Reference ``ContractionTree.slice_and_reconfigure`` (core.py:2723-2808):
        slice towards ``target_size`` ``step_size`` bits at a time, re-optimising
        subtrees in between (``reconf_opts``: ``subtree_size``, ``maxiter``)."""
        from .pathfind import slice_and_reconfigure

        opts = dict(reconf_opts or {})
        new = slice_and_reconfigure(
            self.unslice_all() if reslice else self, target_size, minimize=opts.get("minimize", minimize),
            subtree_size=opts.get("subtree_size", 8), allow_outer=allow_outer, step_bits=float(step_size),
        )
        if inplace:
            self.__dict__.update(new.__dict__)
            return self
        return new

    slice_and_reconfigure_ = functools.partialmethod(slice_and_reconfigure, inplace=True)

    def restore_ind(self, ind, inplace=False):
        """Undo :meth:`remove_ind` (reference core.py:2046-2089)."""
        tree = self if inplace else self.copy()
        si = tree.sliced_inds.pop(ind)
        tree.multiplicity //= si.size
        tree.sliced_inputs = frozenset(
            i
            for i, term in enumerate(tree.inputs)
            if any(ix in tree.sliced_inds for ix in term)
        )
        tree._invalidate()
        return tree

    restore_ind_ = functools.partialmethod(restore_ind, inplace=True)

    def unslice_all(self, inplace=False):
        tree = self if inplace else self.copy()
        for ind in tuple(tree.sliced_inds):
            tree.restore_ind_(ind)
        return tree

    unslice_all_ = functools.partialmethod(unslice_all, inplace=True)

    def slice_key(self, i, strides=None):
        """Value of every sliced index for overall slice ``i`` -- a
        mixed-radix decode, outer indices most significant (reference
        core.py:3775-3800)."""
        if strides is None:
            strides = get_slice_strides(self.sliced_inds)
        key = {}
        for (ind, info), stride in zip(self.sliced_inds.items(), strides):
            if info.project is None:
                key[ind] = i // stride
                i %= stride
            else:
                key[ind] = info.project
        return key

    def slice_arrays(self, arrays, i):
        """Index the sliced inputs at slice ``i`` (views; reference
        core.py:3802-3819).  Works for anything supporting numpy-style
        integer/slice indexing, including torch tensors."""
        out = list(arrays)
        loc = self.slice_key(i)
        for c in self.sliced_inputs:
            selector = tuple(loc.get(ix, slice(None)) for ix in self.inputs[c])
            out[c] = out[c][selector]
        return out

    # ------------------------------------------------------------------ #
    # execution drivers (delegating to the HIP contractor)
    # ------------------------------------------------------------------ #

    def get_contractor(
        self,
        order=None,
        prefer_einsum=False,
        strip_exponent=False,
        check_zero=False,
        implementation=None,
        autojit=False,
        progbar=False,
    ):
        """Cached whole-slice contraction function ``fn(*arrays)`` (reference
        core.py:3638-3722).  ``implementation`` None/"auto"/"hip" selects the
        MI355X executor; there is no CPU fallback."""
        from .contractor import make_contractor

        key = (
            autojit,
            order if not callable(order) else id(order),
            prefer_einsum,
            strip_exponent,
            check_zero,
            implementation
            if not isinstance(implementation, (tuple, list))
            else tuple(map(id, implementation)),
            progbar,
        )
        try:
            fn = self.contraction_cores[key]
        except KeyError:
            fn = self.contraction_cores[key] = make_contractor(
                tree=self,
                order=order,
                prefer_einsum=prefer_einsum,
                strip_exponent=strip_exponent,
                check_zero=check_zero,
                implementation=implementation,
                autojit=autojit,
                progbar=progbar,
            )
        return fn

    def contract_core(
        self,
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
        """Contract already-sliced ``arrays`` (reference core.py:3724-3773).
        """
        if autojit == "auto":
            autojit = False
        fn = self.get_contractor(
            order=order,
            prefer_einsum=prefer_einsum,
            strip_exponent=strip_exponent is not False,
            implementation=implementation,
            autojit=autojit,
            check_zero=check_zero,
            progbar=progbar,
        )
        return fn(*arrays, backend=backend)

    def contract_slice(self, arrays, i, **kwargs):
        """Contract slice ``i`` of the *unsliced* ``arrays`` (reference
        core.py:3821-3823).  On the HIP path slicing is a base-pointer
        offset inside the executor, not a host-side copy."""
        from .contractor import contract_slice

        return contract_slice(self, arrays, i, **kwargs)

    def gather_slices(self, slices, backend=None, progbar=False):
        """Combine per-slice outputs: sum over inner sliced indices, stack
        over outer ones (reference core.py:3825-3882)."""
        from .contractor import gather_slices

        return gather_slices(self, slices, backend=backend, progbar=progbar)

    def gen_output_chunks(
        self, arrays, with_key=False, progbar=False, **contract_opts
    ):
        """Yield each output chunk with inner sliced indices already summed
        (reference core.py:3884-3941)."""
        from .contractor import gen_output_chunks

        yield from gen_output_chunks(
            self, arrays, with_key=with_key, progbar=progbar, **contract_opts
        )

    def contract(
        self,
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
        """Contract the *unsliced* ``arrays``: slicing, per-slice contraction
        and gathering all happen on the device (reference core.py:3943-4030).
        """
        from .contractor import contract_tree

        return contract_tree(
            self,
            arrays,
            order=order,
            prefer_einsum=prefer_einsum,
            strip_exponent=strip_exponent,
            check_zero=check_zero,
            backend=backend,
            implementation=implementation,
            autojit=autojit,
            progbar=progbar,
        )

    def contract_distributed(self, arrays, group=None, root=None, **kwargs):
        """Slice-parallel contraction over the ranks of a
        ``torch.distributed`` process group, one GPU per rank, finished by a
        single RCCL (all-)reduce -- the MI355X counterpart of the reference's
        ``contract_mpi`` (core.py:4032-4090)."""
        from .distributed import contract_distributed

        return contract_distributed(
            self, arrays, group=group, root=root, **kwargs
        )

    def contract_resumable(self, arrays, checkpoint, **kwargs):
        """:meth:`contract` with a checkpoint file: the partial sum over slices
        is saved every ``every`` slices and an interrupted run continues from
        it bit-identically (``cotengra_amd.contractor.contract_resumable``)."""
        from .contractor import contract_resumable

        return contract_resumable(self, arrays, checkpoint, **kwargs)

    def contract_mpi(self, arrays, comm=None, root=None, **kwargs):
        """The reference's name and signature (core.py:4032-4090).  ``comm`` may
        be an mpi4py communicator as there (it only carries RCCL's unique id to
        the other ranks), a ``torch.distributed`` group, a
        ``cotengra_amd.runtime.Comm``, or None for the default torch group;
        ``root=None`` leaves the total on every rank, ``root=r`` only on rank
        ``r`` (the others return None).  ``kwargs`` as for ``contract_slice``:
        ``order``, ``strip_exponent``, ``check_zero``."""
        from .distributed import contract_distributed

        opts = {k: kwargs.pop(k) for k in ("order", "strip_exponent", "check_zero", "progbar") if k in kwargs}
        for k in ("prefer_einsum", "backend", "implementation", "autojit"):
            kwargs.pop(k, None)
        if kwargs:
            raise TypeError(f"Unknown keyword arguments: {kwargs}.")
        return contract_distributed(self, arrays, comm=comm, root=root, **opts)

    def benchmark(
        self,
        dtype="float64",
        max_time=60,
        min_reps=3,
        max_reps=100,
        warmup=True,
        **contract_opts,
    ):
        """Time ``contract_slice`` on synthetic inputs and extrapolate to all
        slices (reference core.py:4092-4164; same keys in the result)."""
        from .contractor import benchmark_tree

        return benchmark_tree(
            self,
            dtype=dtype,
            max_time=max_time,
            min_reps=min_reps,
            max_reps=max_reps,
            warmup=warmup,
            **contract_opts,
        )

    def descend(self, mode="dfs"):
        """Generate ``(parent, left, right)`` merges from the root down, parents before
        their children; ``mode`` "dfs" pops the newest node, "bfs" the oldest (reference
        core.py:1866-1896)."""
        if self.N == 1:
            return
        queue = [self.root]
        while queue:
            if mode == "dfs":
                parent = queue.pop(-1)
            elif mode == "bfs":
                parent = queue.pop(0)
            else:
                raise ValueError(f"Unknown descend mode {mode!r}.")
            l, r = self.children[parent]
            yield parent, l, r
            if not self.is_leaf(l):
                queue.append(l)
            if not self.is_leaf(r):
                queue.append(r)

    def reset_contraction_indices(self):
        """Forget every explicit index order and everything derived from one -- ``inds``,
        ``einsum_eq``, ``can_dot``, ``tensordot_axes``, ``tensordot_perm`` of the parents --
        and the cached contractors (reference core.py:3400-3419).  Legs, involved indices,
        sizes and flops do not depend on an order and stay."""
        for node in self.children:
            d = self._info.get(node)
            if d:
                for k in ("inds", "einsum_eq", "can_dot", "tensordot_axes", "tensordot_perm"):
                    d.pop(k, None)
        self.contraction_cores.clear()

    def sort_contraction_indices(self, priority="flops", make_output_contig=True,
                                 make_contracted_contig=True, reset=True):
        """Set an explicit index order on every intermediate so that contracted indices
        are contiguous in both children of a contraction and / or a parent's indices
        come in the order of its children's (reference core.py:3421-3506; index work,
        reproduced exactly: ``get_inds`` / ``get_tensordot_axes`` / ``get_tensordot_perm``
        / ``get_einsum_eq`` -- hence ``extract_contractions``' IR -- afterwards are the
        reference's, ``tests/golden/sorted_inds_cases.json``).

        Nodes are visited by ``priority`` -- "flops" / "size": ascending (a stable sort
        of the parents in the order they were added, so the costliest contraction is
        handled last and keeps its order), "root": ``traverse()``, "leaves": ``descend()``
        -- and each visit (i) sorts the parent's indices by ``(position in the right
        child, position in the left child)`` with -1 for "absent" (not at the root: the
        output order is the caller's), (ii) sorts the legs of each non-leaf child so that
        the indices it shares with its sibling come last (left child) resp. first (right
        child), ties by position in the parent.  Orders are read through the same lazy
        cache as everywhere else, so a later visit sees what earlier visits set and
        nothing else -- the outcome depends on it and is the reference's.

        The result of a contraction does not depend on any of this, and the MI355X plan
        compiler chooses its own memory layouts (no permutation is ever materialised:
        DESIGN.md section 2); what changes is the tree's reported index algebra, i.e.
        the per-op plug-in's ``tensordot`` axes / ``einsum`` equations and ``transpose``
        calls (``PerOpContractor``) and everything printed from them."""
        if reset:
            self.reset_contraction_indices()

        if priority == "flops":
            nodes = sorted(self.children.items(), key=lambda x: self.get_flops(x[0]))
        elif priority == "size":
            nodes = sorted(self.children.items(), key=lambda x: self.get_size(x[0]))
        elif priority == "root":
            nodes = ((p, (l, r)) for p, l, r in self.traverse())
        elif priority == "leaves":
            nodes = ((p, (l, r)) for p, l, r in self.descend())
        else:
            raise ValueError(priority)

        def find(inds, ix):   # ``str.find`` on a tuple of labels
            try:
                return inds.index(ix)
            except ValueError:
                return -1

        for p, kids in nodes:
            if len(kids) != 2:
                continue   # (a one-input tree has nothing to sort)
            l, r = kids
            p_inds, l_inds, r_inds = map(self.get_inds_tuple, (p, l, r))

            if make_output_contig and not self.is_root(p):
                p_inds = tuple(sorted(p_inds, key=lambda ix: (find(r_inds, ix), find(l_inds, ix))))
                self._info.setdefault(p, {})["inds"] = p_inds

            if make_contracted_contig:
                if not self.is_leaf(l):
                    l_inds = tuple(sorted(self.get_legs(l), key=lambda ix: (find(r_inds, ix), find(p_inds, ix))))
                    self._info.setdefault(l, {})["inds"] = l_inds
                if not self.is_leaf(r):
                    r_inds = tuple(sorted(self.get_legs(r), key=lambda ix: (find(p_inds, ix), find(l_inds, ix))))
                    self._info.setdefault(r, {})["inds"] = r_inds

        if not reset:
            # (still invalidate the compiled contractions)
            self.contraction_cores.clear()

    def print_contractions(self, sort=None, show_brackets=True):
        """Per-step cost table (cf. reference core.py:3508): step, log10
        scalar ops, log2 result size, and the pairwise einsum equation."""
        rows = []
        for i, (p, l, r) in enumerate(self.traverse()):
            rows.append((i, math.log10(max(self.get_flops(p), 1)), math.log2(max(self.get_size(p), 1)),
                         self.get_einsum_eq(p)))
        if sort == "flops":
            rows.sort(key=lambda t: -t[1])
        elif sort == "size":
            rows.sort(key=lambda t: -t[2])
        for i, f, s_, eq in rows:
            print(f"({i}) cost: {f:4.1f} width: {s_:4.1f} {eq if len(eq) < 120 else eq[:117] + '...'}")

    # ------------------------------------------------------------------ #
    # description
    # ------------------------------------------------------------------ #

    def describe(self):
        return (
            f"log10[FLOPs]={self.total_flops(log=10):.2f} "
            f"log2[SIZE]={self.max_size(log=2):.2f} "
            f"nslices={self.nslices}"
        )

    def __repr__(self):
        return (
            f"<{self.__class__.__name__}(N={self.N}, "
            f"sliced={len(self.sliced_inds)}, {self.describe()})>"
        )
