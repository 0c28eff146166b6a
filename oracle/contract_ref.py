"""ORACLE -- test infrastructure, NOT product code.

A numpy restatement of the reference's contraction-tree executor
(cotengra v0.8.2): the bmm lowering of ``cotengra/contract.py`` and the slice
loop / gather of ``cotengra/core.py``.  It exists to (1) check the HIP path
in ``tests/`` and ``__graft_entry__.smoke()`` and (2) serve as the CPU
baseline timed by ``bench.py`` on the GPU node's host cores (``cpu_baseline``,
kind "port").  Nothing under ``cotengra_amd/`` may import it.

Parity status: PINNED.  ``tests/golden/gen/make_golden.py`` imports the real
reference in the build container (through the test-only ``oracle/refshim``
autoray shim) and (a) compares this file's lowering tuples and results with
the reference's on every golden case, (b) writes the reference's own outputs
to ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` re-checks this
file against those committed vectors on every run.

Every function cites the reference lines it follows.
"""

from __future__ import annotations

import functools
import itertools
import operator

import numpy as np

# --------------------------------------------------------------------------- #
# single-term einsum   (reference contract.py:34-119, 332-361)
# --------------------------------------------------------------------------- #


def sanitize_equation(eq):
    """contract.py:34-58 -- split lhs/out, computing the implicit output as
    the sorted indices that appear exactly once."""
    eq = eq.replace(" ", "")
    if "..." in eq:
        raise NotImplementedError("Ellipsis not supported.")
    if "->" not in eq:
        lhs = eq
        flat = lhs.replace(",", "")
        out = "".join(s for s in sorted(set(flat)) if flat.count(s) == 1)
    else:
        lhs, out = eq.split("->")
    return lhs, out


def parse_einsum_single(eq, shape):
    """contract.py:61-119 -- (diagonal selectors, summed axes, permutation)."""
    lhs, out = sanitize_equation(eq)
    need_to_diag, need_to_sum, seen = [], [], set()
    for ix in lhs:
        if ix in need_to_diag:
            continue
        if ix in seen:
            need_to_diag.append(ix)
            continue
        seen.add(ix)
        if ix not in out:
            need_to_sum.append(ix)

    if need_to_diag:
        diag_sels = []
        sizes = dict(zip(lhs, shape))
        while need_to_diag:
            ixd = need_to_diag.pop()
            dinds = tuple(range(sizes[ixd]))
            diag_sels.append(
                tuple(dinds if ix == ixd else slice(None) for ix in lhs)
            )
            contig = ixd * lhs.count(ixd)
            if contig in lhs:
                lhs = lhs.replace(contig, ixd)
            else:
                lhs = ixd + lhs.replace(ixd, "")
    else:
        diag_sels = None

    if need_to_sum:
        sum_axes = tuple(map(lhs.index, need_to_sum))
        for ix in need_to_sum:
            lhs = lhs.replace(ix, "")
    else:
        sum_axes = None

    perm = None if lhs == out else tuple(lhs.index(ix) for ix in out)
    return diag_sels, sum_axes, perm


def einsum_single(eq, x):
    """contract.py:332-361 -- diagonal via advanced indexing, sum, transpose
    (the branch taken when the array library has no ``einsum``)."""
    diag_sels, sum_axes, perm = parse_einsum_single(eq, tuple(x.shape))
    if diag_sels is not None:
        for sel in diag_sels:
            x = x[sel]
    if sum_axes is not None:
        x = np.sum(x, sum_axes)
    if perm is not None:
        x = np.transpose(x, perm)
    return x


# --------------------------------------------------------------------------- #
# pairwise lowering to (batched) matmul   (reference contract.py:122-329)
# --------------------------------------------------------------------------- #


def parse_eq_to_pure_multiplication(a_term, shape_a, b_term, shape_b, out):
    """contract.py:122-164 -- no contracted index: align both operands to the
    output with singleton axes so a broadcast multiply does the einsum."""
    desired_a = desired_b = ""
    new_shape_a, new_shape_b = [], []
    for ix in out:
        if ix in a_term:
            desired_a += ix
            new_shape_a.append(shape_a[a_term.index(ix)])
        else:
            new_shape_a.append(1)
        if ix in b_term:
            desired_b += ix
            new_shape_b.append(shape_b[b_term.index(ix)])
        else:
            new_shape_b.append(1)
    eq_a = f"{a_term}->{desired_a}" if desired_a != a_term else None
    eq_b = f"{b_term}->{desired_b}" if desired_b != b_term else None
    return eq_a, eq_b, new_shape_a, new_shape_b, None, None, True


@functools.lru_cache(2**12)
def parse_eq_to_batch_matmul(eq, shape_a, shape_b):
    """contract.py:167-329 -- classify indices into batch / contracted /
    kept-left / kept-right (size-1 axes dropped up front) and emit
    ``(eq_a, eq_b, new_shape_a, new_shape_b, new_shape_ab, perm_ab,
    pure_multiplication)``."""
    lhs, out = eq.split("->")
    a_term, b_term = lhs.split(",")
    if len(a_term) != len(shape_a):
        raise ValueError(f"Term '{a_term}' does not match shape {shape_a}.")
    if len(b_term) != len(shape_b):
        raise ValueError(f"Term '{b_term}' does not match shape {shape_b}.")

    sizes, singletons = {}, set()
    left = {}
    for ix, d in zip(a_term, shape_a):
        if d == 1:
            singletons.add(ix)
            continue
        if sizes.setdefault(ix, d) != d:
            raise ValueError(
                f"Index {ix} has mismatched sizes {sizes[ix]} and {d}."
            )
        left[ix] = True
    right = {}
    for ix, d in zip(b_term, shape_b):
        if d == 1:
            if ix not in left:
                singletons.add(ix)
            continue
        singletons.discard(ix)
        if sizes.setdefault(ix, d) != d:
            raise ValueError(
                f"Index {ix} has mismatched sizes {sizes[ix]} and {d}."
            )
        right[ix] = True

    bat, con, a_keep, b_keep = [], [], [], []
    for ix in left:
        if right.pop(ix, False):
            (bat if ix in out else con).append(ix)
        elif ix in out:
            a_keep.append(ix)
    for ix in right:
        if ix in out:
            b_keep.append(ix)

    if not con:
        return parse_eq_to_pure_multiplication(
            a_term, shape_a, b_term, shape_b, out
        )

    singletons = [ix for ix in out if ix in singletons]

    def prep(term, desired):
        if term == desired:
            return None
        if set(term) == set(desired):
            return tuple(term.index(ix) for ix in desired)
        return f"{term}->{desired}"

    eq_a = prep(a_term, "".join((*bat, *a_keep, *con)))
    eq_b = prep(b_term, "".join((*bat, *con, *b_keep)))

    if bat:
        lgroups, rgroups = (bat, a_keep, con), (bat, con, b_keep)
        ogroups = (bat, a_keep, b_keep)
    else:
        lgroups, rgroups, ogroups = (a_keep, con), (con, b_keep), (a_keep, b_keep)

    def fused(groups):
        if any(len(g) != 1 for g in groups):
            return tuple(
                functools.reduce(operator.mul, (sizes[ix] for ix in g), 1)
                for g in groups
            )
        return None

    new_shape_a, new_shape_b = fused(lgroups), fused(rgroups)
    if any(len(g) != 1 for g in ogroups) or singletons:
        new_shape_ab = (1,) * len(singletons) + tuple(
            sizes[ix] for g in ogroups for ix in g
        )
    else:
        new_shape_ab = None

    produced = "".join((*singletons, *bat, *a_keep, *b_keep))
    perm_ab = (
        tuple(produced.index(ix) for ix in out) if produced != out else None
    )
    return eq_a, eq_b, new_shape_a, new_shape_b, new_shape_ab, perm_ab, False


def do_contraction_via_bmm(
    a, b, eq_a, eq_b, new_shape_a, new_shape_b, new_shape_ab, perm_ab, pure
):
    """contract.py:364-411 -- transpose/single-einsum, reshape, matmul (or
    multiply), reshape, transpose."""
    if eq_a is not None:
        a = np.transpose(a, eq_a) if isinstance(eq_a, tuple) else einsum_single(eq_a, a)
    if new_shape_a is not None:
        a = np.reshape(a, new_shape_a)
    if eq_b is not None:
        b = np.transpose(b, eq_b) if isinstance(eq_b, tuple) else einsum_single(eq_b, b)
    if new_shape_b is not None:
        b = np.reshape(b, new_shape_b)
    if pure:
        return np.multiply(a, b)
    ab = np.matmul(a, b)
    if new_shape_ab is not None:
        ab = np.reshape(ab, new_shape_ab)
    if perm_ab is not None:
        ab = np.transpose(ab, perm_ab)
    return ab


def einsum(eq, a, b=None):
    """contract.py:414-459."""
    if b is None:
        return einsum_single(eq, a)
    parsed = parse_eq_to_batch_matmul(eq, tuple(a.shape), tuple(b.shape))
    return do_contraction_via_bmm(a, b, *parsed)


def _nice_inds():
    """contract.py:462-469."""
    for i in range(26):
        yield chr(ord("a") + i)
    for i in range(26):
        yield chr(ord("A") + i)
    for i in itertools.count(192):
        yield chr(i)


@functools.lru_cache(2**12)
def parse_tensordot_axes_to_matmul(axes, shape_a, shape_b):
    """contract.py:472-518 -- turn tensordot ``axes`` into an einsum eq."""
    ndim_a, ndim_b = len(shape_a), len(shape_b)
    if isinstance(axes, int):
        axes_a = tuple(range(ndim_a - axes, ndim_a))
        axes_b = tuple(range(axes))
    else:
        axes_a, axes_b = axes
    if len(axes_a) != len(axes_b):
        raise ValueError(
            f"Axes should have the same length, got {axes_a} and {axes_b}."
        )
    gen = _nice_inds()
    inds_a = [next(gen) for _ in range(ndim_a)]
    inds_b = []
    inds_out = inds_a.copy()
    for axb in range(ndim_b):
        if axb not in axes_b:
            ind = next(gen)
            inds_out.append(ind)
        else:
            axa = axes_a[axes_b.index(axb)]
            if shape_a[axa] != shape_b[axb]:
                raise ValueError(
                    f"Dimension mismatch between axes {axa} of {shape_a} and "
                    f"{axb} of {shape_b}: {shape_a[axa]} != {shape_b[axb]}."
                )
            ind = inds_a[axa]
            inds_out.remove(ind)
        inds_b.append(ind)
    eq = f"{''.join(inds_a)},{''.join(inds_b)}->{''.join(inds_out)}"
    return parse_eq_to_batch_matmul(eq, shape_a, shape_b)


def tensordot(a, b, axes=2):
    """contract.py:521-570."""
    try:
        axes = tuple(map(int, axes[0])), tuple(map(int, axes[1]))
    except (IndexError, TypeError):
        axes = int(axes)
    parsed = parse_tensordot_axes_to_matmul(
        axes, tuple(a.shape), tuple(b.shape)
    )
    return do_contraction_via_bmm(a, b, *parsed)


# --------------------------------------------------------------------------- #
# tree -> op list -> execution   (reference contract.py:573-651, 718-837)
# --------------------------------------------------------------------------- #


def extract_contractions(tree, order=None, prefer_einsum=False):
    """contract.py:573-651 -- the linear IR ``(p, l, r, tdot, arg, perm)`` in
    SSA ids; works on any object with the tree metadata interface."""
    if tree.N == 1:
        return [(1, 0, None, False, tree.get_eq_sliced(), None)]
    contractions = []
    ssas = {leaf: i for i, leaf in enumerate(tree.gen_leaves())}
    ssa = len(ssas)
    for p, l, r in tree.traverse(order=order):
        li, ri = ssas.pop(l), ssas.pop(r)
        pi = ssas[p] = ssa
        ssa += 1
        if prefer_einsum or not tree.get_can_dot(p):
            tdot, arg, perm = False, tree.get_einsum_eq(p), None
        else:
            tdot = True
            arg = tree.get_tensordot_axes(p)
            perm = tree.get_tensordot_perm(p)
        contractions.append((pi, li, ri, tdot, arg, perm))
    if tree.preprocessing:
        pre = ((i, None, None, False, eq, None) for i, eq in tree.preprocessing.items())
        return (*pre, *contractions)
    return tuple(contractions)


def run_contractions(contractions, arrays, strip_exponent=False, check_zero=False):
    """contract.py:718-837 (``Contractor.__call__`` with the "cotengra"
    implementation): run the IR over a dict of temporaries, popping operands,
    optionally stripping the base-10 exponent after every step."""
    temps = dict(enumerate(arrays))
    exponent = 0.0 if strip_exponent else None
    p_array = None
    for pi, li, ri, tdot, arg, perm in contractions:
        if ri is None:
            if li is None:
                temps[pi] = einsum(arg, temps[pi])
                continue
            p_array = einsum(arg, temps[li])
            if strip_exponent:
                return p_array, 0.0
            return p_array
        l_array, r_array = temps.pop(li), temps.pop(ri)
        if tdot:
            p_array = tensordot(l_array, r_array, arg)
            if perm:
                p_array = np.transpose(p_array, perm)
        else:
            p_array = einsum(arg, l_array, r_array)
        if exponent is not None:
            factor = np.max(np.abs(p_array))
            if check_zero and float(factor) == 0.0:
                return 0.0, float("-inf")
            exponent = exponent + np.log10(factor)
            p_array = p_array / factor
        temps[pi] = p_array
    if exponent is not None:
        return p_array, exponent
    return p_array


# --------------------------------------------------------------------------- #
# slicing and gathering   (reference core.py:114-172, 3775-3941, 3943-4030)
# --------------------------------------------------------------------------- #


def slice_strides(sizes):
    """core.py:114-122."""
    strides = [1] * len(sizes)
    for i in range(len(sizes) - 2, -1, -1):
        strides[i] = strides[i + 1] * sizes[i + 1]
    return strides


def slice_key(tree, i):
    """core.py:3775-3800."""
    infos = list(tree.sliced_inds.values())
    strides = slice_strides([si.size for si in infos])
    key = {}
    for info, stride in zip(infos, strides):
        if info.project is None:
            key[info.ind] = i // stride
            i %= stride
        else:
            key[info.ind] = info.project
    return key


def slice_arrays(tree, arrays, i):
    """core.py:3802-3819."""
    temp = list(arrays)
    loc = slice_key(tree, i)
    for c in tree.sliced_inputs:
        sel = tuple(loc.get(ix, slice(None)) for ix in tree.inputs[c])
        temp[c] = temp[c][sel]
    return temp


def add_maybe_exponent_stripped(x, y):
    """core.py:125-172."""
    xt, yt = isinstance(x, tuple), isinstance(y, tuple)
    if not (xt or yt):
        return x + y
    xm, xe = x if xt else (x, 0.0)
    ym, ye = y if yt else (y, 0.0)
    e = max(xe, ye)
    return xm * 10 ** (xe - e) + ym * 10 ** (ye - e), e


def gather_slices(tree, slices):
    """core.py:3825-3882 -- sum over inner sliced indices, stack over outer."""
    output_pos = {
        ix: i for i, ix in enumerate(tree.output) if ix in tree.sliced_inds
    }
    if not output_pos:
        return functools.reduce(add_maybe_exponent_stripped, slices)
    chunks = {}
    for i, s in enumerate(slices):
        ks = slice_key(tree, i)
        key = tuple(ks[ix] for ix in output_pos)
        chunks[key] = (
            add_maybe_exponent_stripped(chunks[key], s) if key in chunks else s
        )
    if isinstance(next(iter(chunks.values())), tuple):
        emax = max(v[1] for v in chunks.values())
        chunks = {k: m * 10 ** (e - emax) for k, (m, e) in chunks.items()}
    else:
        emax = None

    def stack(loc, remaining):
        if not remaining:
            return chunks[loc]
        arrs = [
            stack(loc + (d,), remaining[1:])
            for d in tree.sliced_inds[remaining[0]].sliced_range
        ]
        return np.stack(arrs, output_pos[remaining[0]] - len(loc))

    result = stack((), tuple(output_pos))
    return (result, emax) if emax is not None else result


def contract_slice(tree, arrays, i, **opts):
    """core.py:3821-3823."""
    ops = extract_contractions(
        tree, opts.pop("order", None), opts.pop("prefer_einsum", False)
    )
    return run_contractions(ops, slice_arrays(tree, arrays, i), **opts)


def contract(tree, arrays, order=None, prefer_einsum=False, strip_exponent=False,
             check_zero=False):
    """core.py:3943-4030."""
    ops = extract_contractions(tree, order, prefer_einsum)
    kw = dict(strip_exponent=strip_exponent, check_zero=check_zero)
    if not tree.sliced_inds:
        return run_contractions(ops, arrays, **kw)
    slices = (
        run_contractions(ops, slice_arrays(tree, arrays, i), **kw)
        for i in range(tree.multiplicity)
    )
    return gather_slices(tree, slices)
