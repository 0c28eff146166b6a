"""Small host-side helpers: index symbols, equation strings, and the
synthetic workload/array generators the benchmarks are defined on.

Only what the execution path and its tests need; each function cites the
reference behaviour it reproduces (cotengra v0.8.2, ``cotengra/utils.py``).
"""

from __future__ import annotations

import functools
import itertools
import json
import operator

_SYMBOLS_BASE = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"


def prod(it):
    return functools.reduce(operator.mul, it, 1)


def unique(it):
    """Order-preserving de-duplication."""
    return dict.fromkeys(it).keys()


@functools.lru_cache(2**14)
def get_symbol(i):
    """``i``-th index symbol: a-z, A-Z, then unicode from chr(192) skipping
    the surrogate block (reference utils.py:657-688)."""
    if i < 52:
        return _SYMBOLS_BASE[i]
    i += 140
    if i >= 55296:
        i += 2048
    return chr(i)


def inputs_output_to_eq(inputs, output, canonicalize=False):
    """Explicit terms -> einsum string; with ``canonicalize`` labels are
    renamed a, b, c... in order of first appearance (reference
    utils.py:1145-1170)."""
    if canonicalize:
        table = {}

        def rename(ix):
            try:
                return table[ix]
            except KeyError:
                s = table[ix] = get_symbol(len(table))
                return s

        inputs = [[rename(ix) for ix in term] for term in inputs]
        output = [rename(ix) for ix in output]
    return f"{','.join(''.join(t) for t in inputs)}->{''.join(output)}"


def find_output_str(lhs):
    """Implicit einsum output: indices appearing exactly once, sorted
    (reference utils.py:1100-1117)."""
    flat = lhs.replace(",", "")
    return "".join(s for s in sorted(set(flat)) if flat.count(s) == 1)


def eq_to_inputs_output(eq):
    """Einsum string -> (inputs, output) tuples of single-character labels
    (reference utils.py:1120-1142)."""
    eq = eq.replace(" ", "")
    lhs, *rhs = eq.split("->")
    inputs = tuple(map(tuple, lhs.split(",")))
    output = tuple(rhs[0]) if rhs else tuple(find_output_str(lhs))
    return inputs, output


def shapes_inputs_to_size_dict(shapes, inputs):
    """Index sizes from matching shapes / terms; a size-1 axis never
    overrides a larger extent (broadcast) but two different extents > 1 are
    an error."""
    size_dict = {}
    for term, shape in zip(inputs, shapes):
        if len(term) != len(shape):
            raise ValueError(f"Term {term} does not match shape {shape}.")
        for ix, d in zip(term, shape):
            d = int(d)
            old = size_dict.get(ix)
            if old is None or old == 1:
                size_dict[ix] = d
            elif d != 1 and d != old:
                raise ValueError(
                    f"Index {ix} has mismatched sizes {old} and {d}."
                )
    return size_dict


def lattice_equation(dims, cyclic=False, d_min=2, d_max=None, seed=None):
    """Hyper-cubic lattice network: one tensor per site, one bond per
    nearest-neighbour pair, no output (reference utils.py:1028-1096).  Bond
    symbols are assigned in order of first appearance while sites are visited
    in row-major order, each looking at its -1 then +1 neighbour per axis.
    Returns ``(inputs, output, shapes, size_dict)``.
    """
    import random

    if d_max is None:
        d_max = d_min
    ndim = len(dims)
    try:
        cyclics = tuple(cyclic)
    except TypeError:
        cyclics = (cyclic,) * ndim

    symbols = {}
    inputs = []
    for site in itertools.product(*(range(n) for n in dims)):
        term = []
        for axis in range(ndim):
            for step in (-1, +1):
                other = list(site)
                other[axis] += step
                if cyclics[axis]:
                    other[axis] %= dims[axis]
                elif not (0 <= other[axis] < dims[axis]):
                    continue
                other = tuple(other)
                edge = (site, other) if site < other else (other, site)
                if edge not in symbols:
                    symbols[edge] = get_symbol(len(symbols))
                term.append(symbols[edge])
        inputs.append(term)

    rng = random.Random(seed)
    size_dict = {ix: int(rng.randint(d_min, d_max)) for ix in symbols.values()}
    shapes = tuple(tuple(size_dict[ix] for ix in term) for term in inputs)
    return inputs, [], shapes, size_dict


def make_arrays_from_inputs(
    inputs, size_dict, seed=None, dtype="float64", rescale=False
):
    """Synthetic input tensors with the reference's exact draw order
    (reference utils.py:1243-1284): one ``default_rng(seed)``; per tensor a
    standard-normal real part, then (complex dtypes) a standard-normal
    imaginary part; cast; divide by the Frobenius norm.

    ``rescale=True`` additionally multiplies each tensor by ``size**0.25`` so
    that a closed network evaluates to O(1) instead of underflowing fp32
     -- applied identically to oracle and device inputs.
    """
    import numpy as np

    rng = np.random.default_rng(seed)
    arrays = []
    for term in inputs:
        shape = tuple(size_dict[ix] for ix in term)
        x = rng.normal(size=shape)
        if dtype == "float32":
            x = x.astype(np.float32)
        elif dtype == "complex64":
            x = (x + 1j * rng.normal(size=shape)).astype(np.complex64)
        elif dtype == "complex128":
            x = x + 1j * rng.normal(size=shape)
        elif dtype != "float64":
            raise ValueError(f"unsupported dtype {dtype}")
        x /= np.linalg.norm(x)
        if rescale:
            x *= x.dtype.type(float(x.size) ** 0.25)
        arrays.append(x)
    return arrays


def load_network(filename):
    """Load ``{inputs, output, size_dict[, path, sliced_inds]}`` JSON -- the
    reference's benchmark/persistence format (utils.py:1628-1650,
    hyperoptimizers/hyper.py:1075-1096)."""
    with open(filename, "r", encoding="utf-8") as f:
        data = json.load(f)
    data["inputs"] = [tuple(t) for t in data["inputs"]]
    data["output"] = tuple(data["output"])
    return data


def tree_from_record(rec):
    """Rebuild a (possibly sliced) tree from a ``{path, sliced_inds}`` record.
    """
    from .tree import ContractionTree

    tree = ContractionTree.from_path(
        rec["inputs"], rec["output"], rec["size_dict"], path=rec["path"]
    )
    # entries are index names, or [name, j] for an index projected onto value j
    return tree.apply_slicing_(rec.get("sliced_inds", ()))
