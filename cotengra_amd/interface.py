"""einsum-style front ends over the HIP executor.

Mirrors the call shapes of the reference's high level API
(``cotengra/interface.py``): ``einsum`` (:1038), ``einsum_tree`` (:875),
``einsum_expression`` (:925), ``array_contract`` (:803),
``array_contract_tree`` (:394), ``array_contract_expression`` (:673).

The reference's hyper-optimizers stay on the host unchanged and hand over a
path or a tree.  ``optimize`` therefore accepts an explicit
path (list of pairs), a ``ContractionTree`` (ours, or any object exposing
``inputs/output/size_dict/get_path()/sliced_inds`` such as the reference's
tree), or the string ``"greedy"`` for the native greedy finder that makes the
front ends usable stand-alone (``cotengra_amd.pathfind`` has the rest: slicing,
subtree reconfiguration, ``search`` -- their results are trees to pass here).
"""

from __future__ import annotations

import collections
import os
import threading

import numpy as np

from .contractor import HipContractor, _is_torch
from .tree import ContractionTree
from .utils import eq_to_inputs_output, prod, shapes_inputs_to_size_dict


def greedy_path(inputs, output, size_dict):
    """A plain greedy pairwise path (linear recycled ids): repeatedly contract
    the pair of index-sharing tensors that minimises
    ``size(result) - size(a) - size(b)``; leftover disconnected components are
    combined smallest first.  Hyper-indices and output indices are kept until
    their last appearance.  This is a convenience, not a port of the
    reference's optimizers (``cotengra/pathfinders``)."""
    terms = [frozenset(t) for t in inputs]
    counts = {}
    for t in list(terms) + [frozenset(output)]:
        for ix in t:
            counts[ix] = counts.get(ix, 0) + 1
    # multiplicity-aware appearance count (repeated index on one term counts once
    # here; the tree's preprocessing handles diagonals)
    live = list(range(len(terms)))
    tens = {i: terms[i] for i in live}
    path = []

    def size(t):
        return prod(size_dict[ix] for ix in t)

    def merged(a, b):
        both = a | b
        out = set()
        for ix in both:
            n = (ix in a) + (ix in b)
            if counts[ix] - n > 0:
                out.add(ix)
        return frozenset(out)

    next_id = len(terms)
    while len(live) > 1:
        best = None
        for pi in range(len(live)):
            for pj in range(pi + 1, len(live)):
                a, b = tens[live[pi]], tens[live[pj]]
                if not (a & b):
                    continue
                m = merged(a, b)
                score = size(m) - size(a) - size(b)
                if best is None or score < best[0]:
                    best = (score, pi, pj, m)
        if best is None:
            # disconnected: outer product of the two smallest
            order = sorted(range(len(live)), key=lambda p: size(tens[live[p]]))
            pi, pj = sorted(order[:2])
            m = merged(tens[live[pi]], tens[live[pj]])
        else:
            _, pi, pj, m = best
        a, b = tens[live[pi]], tens[live[pj]]
        for ix in a | b:
            counts[ix] -= (ix in a) + (ix in b) - (ix in m)
        path.append((pi, pj))
        live.pop(pj)
        live.pop(pi)
        tens[next_id] = m
        live.append(next_id)
        next_id += 1
    return tuple(path)


def _as_tree(inputs, output, size_dict, optimize):
    if isinstance(optimize, ContractionTree):
        return optimize
    if isinstance(optimize, str):
        if optimize not in ("greedy", "auto"):
            raise ValueError(
                f"optimize={optimize!r}: only 'greedy' is built in; pass a path "
                "or a tree found by cotengra's optimizers."
            )
        if len(inputs) == 1:
            return ContractionTree(inputs, output, size_dict)
        # the native greedy finder (csrc/ctg_pathfind.cpp); the small Python
        # heuristic above remains for callers that must not load the library
        from .pathfind import greedy_tree

        return greedy_tree(inputs, output, size_dict)
    if hasattr(optimize, "get_path") and hasattr(optimize, "sliced_inds"):
        # a foreign (e.g. reference cotengra) tree: adopt its path and slicing
        tree = ContractionTree.from_path(
            optimize.inputs, optimize.output, optimize.size_dict,
            path=optimize.get_path(),
        )
        # (a mapping ind -> SliceInfo in the reference too: projections carry over)
        return tree.apply_slicing_(optimize.sliced_inds)
    if len(inputs) == 1:
        return ContractionTree(inputs, output, size_dict)
    return ContractionTree.from_path(inputs, output, size_dict, path=optimize)


def array_contract_tree(inputs, output, size_dict, optimize="greedy", sort_contraction_indices=False):
    """interface.py:394 -- build the tree for explicit index lists;
    ``sort_contraction_indices`` calls the tree's method of that name on the result
    (interface.py:455-456)."""
    tree = _as_tree(
        [tuple(t) for t in inputs], tuple(output), dict(size_dict), optimize
    )
    if sort_contraction_indices:
        tree.sort_contraction_indices()
    return tree


def einsum_tree(eq, *shapes, optimize="greedy", sort_contraction_indices=False):
    """interface.py:875."""
    inputs, output = eq_to_inputs_output(eq)
    size_dict = shapes_inputs_to_size_dict(shapes, inputs)
    return array_contract_tree(inputs, output, size_dict, optimize, sort_contraction_indices)


class Via:
    """``fn`` wrapped with input / output conversions -- the reference's
    host<->device hook (interface.py:476-491), e.g.
    ``Via(expr, convert_in=torch.as_tensor, convert_out=lambda x: x.cpu().numpy())``."""

    __slots__ = ("fn", "convert_in", "convert_out")

    def __init__(self, fn, convert_in, convert_out):
        self.fn = fn
        self.convert_in = convert_in
        self.convert_out = convert_out

    def __call__(self, *arrays, **kwargs):
        arrays = map(self.convert_in, arrays)
        out = self.fn(*arrays, **kwargs)
        return self.convert_out(out)


class Variadic:
    """``fn(arrays, **kw)`` exposed as ``fn(*arrays, **kw)`` (interface.py:461-473)."""

    __slots__ = ("fn", "kwargs")

    def __init__(self, fn, **kwargs):
        self.fn = fn
        self.kwargs = kwargs

    def __call__(self, *arrays, **kwargs):
        return self.fn(arrays, **self.kwargs, **kwargs)


class ContractExpression:
    """Reusable ``expr(*arrays)`` (the object ``_build_expression`` returns,
    interface.py:585-667); sliced trees run all their slices on the device."""

    def __init__(self, tree, strip_exponent=False, check_zero=False):
        self.tree = tree
        self.fn = HipContractor(
            tree, strip_exponent=strip_exponent, check_zero=check_zero,
            handle_slicing=True,
        )
        self._cached = False   # set by _cached_expression
        self._bytes = 0        # device memory of its executors as of its last call

    def __call__(self, *arrays, backend=None, **kwargs):
        out = self.fn(*arrays, **kwargs)
        if self._cached:
            self._bytes = self.device_bytes()   # (its executors exist now: the cache's byte bound sees them)
            with _EXPR_LOCK:
                victims = _trim_expression_cache(keep=self)
            _close_all(victims)                 # outside the cache lock, see _trim_expression_cache
        return out

    def device_bytes(self):
        """Device memory this expression's executors hold right now (``ctg_exec_device_bytes``:
        arena x slice batch, inputs, tables, result, scratch), read under the contractor's lock
        when it is free and left at the last known value when another thread is inside."""
        fn = self.fn
        if not fn._lock.acquire(blocking=False):
            return self._bytes
        try:
            n = 0
            for st in fn._execs.values():
                n += st["exec"].device_bytes()
                res = st.get("result")
                if res is not None:   # (a torch-owned result buffer)
                    n += res.numel() * res.element_size()
            return n
        finally:
            fn._lock.release()

    def close(self):
        self.fn.close()


# Expressions built by the one-shot front ends (``einsum``, ``array_contract``,
# ``tensordot``) are kept, least recently used first out: a repeated call with
# the same equation and shapes -- the per-op ``implementation=(einsum,
# tensordot)`` use -- finds its plan, tables and device buffers in place instead
# of rebuilding them (the reference memoises its parsers the same way,
# contract.py:34, 61, 121, 167: ``lru_cache(2**12)``).
# The cache is bounded twice: by count and by the device memory its executors hold
# (inputs, arena, result, tables of every plan an expression has built -- one-shot calls
# over many shapes, the per-op plug-in's pattern, must not pin the HBM the next big tree
# needs), and the executor's out-of-memory retry drops it altogether
# (``evict_expression_cache``).  It is shared by threads: a lock guards the dictionary,
# every expression's contractor serialises its own upload -> run -> fetch.
_EXPR_CACHE = collections.OrderedDict()
_EXPR_CACHE_SIZE = 64
_EXPR_CACHE_BYTES = int(os.environ.get("CTG_EXPR_CACHE_BYTES", 4 << 30))
_EXPR_LOCK = threading.RLock()


def _cached_expression(inputs, output, size_dict, optimize, strip_exponent, check_zero):
    try:
        opt_key = optimize if isinstance(optimize, str) else tuple(map(tuple, optimize))
        key = (tuple(inputs), tuple(output), tuple(sorted(size_dict.items())), opt_key,
               bool(strip_exponent), bool(check_zero))
        hash(key)
    except TypeError:
        key = None  # a tree object or something unhashable: the caller keeps it
    with _EXPR_LOCK:
        if key is not None and key in _EXPR_CACHE:
            _EXPR_CACHE.move_to_end(key)
            return _EXPR_CACHE[key]
        tree = array_contract_tree(inputs, output, size_dict, optimize)
        expr = ContractExpression(tree, strip_exponent, check_zero)
        victims = []
        if key is not None:
            expr._cached = True
            _EXPR_CACHE[key] = expr
            victims = _trim_expression_cache(keep=expr)
    _close_all(victims)
    return expr


def _trim_expression_cache(keep=None):
    """Least recently used out until the cache is within its count and its bytes.  Called
    with ``_EXPR_LOCK`` held; RETURNS the evicted expressions, which the caller closes after
    releasing the lock (``_close_all``): closing waits for the expression's contractor lock,
    and a thread that holds a contractor lock may be waiting for ``_EXPR_LOCK`` in the
    out-of-memory path (``evict_expression_cache``) -- never both at once.  The byte total is
    the sum of every expression's last known size (``ContractExpression._bytes``, refreshed
    by its own calls): no ctypes calls, no foreign locks here."""
    victims = []
    total = sum(e._bytes for e in _EXPR_CACHE.values())
    while len(_EXPR_CACHE) > 1 and (len(_EXPR_CACHE) > _EXPR_CACHE_SIZE or total > _EXPR_CACHE_BYTES):
        k, old = next(iter(_EXPR_CACHE.items()))
        if old is keep:
            break
        del _EXPR_CACHE[k]
        total -= old._bytes
        victims.append(old)
    return victims


def _close_all(exprs):
    """Free the device memory of evicted expressions (no cache lock held)."""
    for old in exprs:
        old.close()


def clear_expression_cache():
    """Close and drop every cached one-shot expression."""
    with _EXPR_LOCK:
        victims = list(_EXPR_CACHE.values())
        _EXPR_CACHE.clear()
    _close_all(victims)


def evict_expression_cache(keep=None):
    """Out-of-memory path of an executor (contractor._get_exec, which holds ITS contractor's
    lock): drop every cached expression except the one whose contractor is ``keep`` (it is
    being built right now) and release the idle executors of the dropped ones without
    waiting for anybody's lock (``_release_execs``: an expression another thread is inside
    of keeps its executor and simply leaves the cache).  True if device memory was released."""
    from .contractor import _release_execs

    with _EXPR_LOCK:
        victims = [_EXPR_CACHE.pop(k) for k in [k for k, e in _EXPR_CACHE.items() if e.fn is not keep]]
    freed = False
    for old in victims:
        freed = _release_execs(old.fn) or freed
    return freed


def array_contract_expression(
    inputs, output, size_dict=None, shapes=None, optimize="greedy",
    strip_exponent=False, check_zero=False, via=None, sort_contraction_indices=False,
    cache_expression=False, **_ignored,
):
    """interface.py:673.  ``via=(convert_in, convert_out)`` wraps the
    expression like the reference does (interface.py:664-665).
    ``sort_contraction_indices`` (interface.py:455-456) sorts the tree's index orders as
    the reference does (``ContractionTree.sort_contraction_indices``: the tree's reported
    index algebra changes, the value does not -- the plan compiler chooses its own memory
    layouts either way); sorted expressions are not shared through the one-shot cache."""
    inputs = [tuple(t) for t in inputs]
    if size_dict is None:
        size_dict = shapes_inputs_to_size_dict(shapes, inputs)
    if cache_expression and not sort_contraction_indices:
        expr = _cached_expression(inputs, output, size_dict, optimize, strip_exponent, check_zero)
    else:
        tree = array_contract_tree(inputs, output, size_dict, optimize, sort_contraction_indices)
        expr = ContractExpression(tree, strip_exponent, check_zero)
    if via is not None:
        expr = Via(expr, *via)
    return expr


def einsum_expression(eq, *shapes, optimize="greedy", **kwargs):
    """interface.py:925."""
    inputs, output = eq_to_inputs_output(eq)
    return array_contract_expression(
        inputs, output, shapes=shapes, optimize=optimize, **kwargs
    )


def array_contract(arrays, inputs, output, optimize="greedy", **kwargs):
    """interface.py:803."""
    shapes = [tuple(x.shape) for x in arrays]
    kwargs.setdefault("cache_expression", True)
    expr = array_contract_expression(
        inputs, output, shapes=shapes, optimize=optimize, **kwargs
    )
    return expr(*arrays)


def tensordot(a, b, axes=2, **kwargs):
    """Pairwise ``tensordot`` on the MI355X with numpy semantics -- the
    second callable of the reference's per-op plug-in pair
    ``implementation=(einsum, tensordot)`` (contract.py:521-570, 775-776):
    output axes are ``[free-a..., free-b...]``."""
    nda, ndb = len(a.shape), len(b.shape)
    try:
        axes_a, axes_b = tuple(map(int, axes[0])), tuple(map(int, axes[1]))
    except (TypeError, IndexError):
        k = int(axes)
        axes_a, axes_b = tuple(range(nda - k, nda)), tuple(range(k))
    if len(axes_a) != len(axes_b):
        raise ValueError(f"Axes should have the same length, got {axes_a} and {axes_b}.")
    axes_a = tuple(x % nda for x in axes_a)
    axes_b = tuple(x % ndb for x in axes_b)
    from .utils import get_symbol

    ia = [get_symbol(i) for i in range(nda)]
    ib, nxt = [], nda
    for j in range(ndb):
        if j in axes_b:
            x = axes_a[axes_b.index(j)]
            if a.shape[x] != b.shape[j]:
                raise ValueError(
                    f"Dimension mismatch between axes {x} of {tuple(a.shape)} and {j} of "
                    f"{tuple(b.shape)}: {a.shape[x]} != {b.shape[j]}."
                )
            ib.append(ia[x])
        else:
            ib.append(get_symbol(nxt))
            nxt += 1
    out = [s for i, s in enumerate(ia) if i not in axes_a] + [s for j, s in enumerate(ib) if j not in axes_b]
    eq = f"{''.join(ia)},{''.join(ib)}->{''.join(out)}"
    return einsum(eq, a, b, optimize=[(0, 1)], **kwargs)


def einsum(eq, *arrays, optimize="greedy", **kwargs):
    """interface.py:1038 -- ``einsum(eq, *arrays)`` on the MI355X."""
    inputs, output = eq_to_inputs_output(eq)
    return array_contract(arrays, inputs, output, optimize=optimize, **kwargs)
