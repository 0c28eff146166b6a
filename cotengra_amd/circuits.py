"""qsim circuit -> amplitude tensor network (the data format on the caller's
side of the contraction path).

The reference ships Sycamore circuits only as qsim gate lists
(``examples/circuit_n53_m*_s0_e0_pABCDCDAB.qsim``) and relies on quimb (absent
here) to turn them into networks.  This module does that step: parse the gate
list, build the network of ``<b| U_T ... U_1 |0>``, and absorb every rank-1 and
rank-2 tensor into a neighbour -- the simplification that leaves the Sycamore
networks with only rank-3/4 tensors (the reference's own m20 benchmark JSON has
365 rank-4 + 16 rank-3 tensors).

Gate set of the Sycamore supremacy circuits (qsim names):
``x_1_2`` = sqrt(X), ``y_1_2`` = sqrt(Y), ``hz_1_2`` = sqrt(W) with
W = (X+Y)/sqrt(2), ``rz(theta)``, ``fs(theta, phi)`` (fSim).
"""

from __future__ import annotations

import numpy as np

from .utils import get_symbol


def parse_qsim(text):
    """Return ``(n_qubits, [(name, qubits, params), ...])`` in file order."""
    lines = [ln.split() for ln in text.strip().splitlines() if ln.strip()]
    n = int(lines[0][0])
    gates = []
    for tok in lines[1:]:
        name = tok[1]
        nq = 2 if name in ("fs", "cz", "is") else 1
        qubits = tuple(int(q) for q in tok[2 : 2 + nq])
        params = tuple(float(x) for x in tok[2 + nq :])
        gates.append((name, qubits, params))
    return n, gates


def gate_matrix(name, params=()):
    """Unitary of a gate, 2x2 or 4x4 (row = output basis state)."""
    s = 1.0 / np.sqrt(2.0)
    if name == "x_1_2":
        return s * np.array([[1, -1j], [-1j, 1]])
    if name == "y_1_2":
        return s * np.array([[1, -1], [1, 1]], dtype=complex)
    if name == "hz_1_2":
        return s * np.array([[1, -np.exp(0.25j * np.pi)], [np.exp(-0.25j * np.pi), 1]])
    if name == "rz":
        (theta,) = params
        return np.diag([np.exp(-0.5j * theta), np.exp(0.5j * theta)])
    if name == "fs":
        theta, phi = params
        c, sn = np.cos(theta), np.sin(theta)
        return np.array(
            [[1, 0, 0, 0], [0, c, -1j * sn, 0], [0, -1j * sn, c, 0], [0, 0, 0, np.exp(-1j * phi)]]
        )
    if name == "cz":
        return np.diag([1, 1, 1, -1]).astype(complex)
    raise ValueError(f"unknown gate {name!r}")


def circuit_to_network(n, gates, bitstring=None, simplify=True, dtype="complex128"):
    """Amplitude network ``<bitstring| circuit |0...0>``.

    ``bitstring`` holds one character per qubit: ``'0'`` / ``'1'`` project the
    qubit, any other character (e.g. ``'?'``) leaves it *open* -- the network
    then has one output index per open qubit (in qubit order) and contracts to
    the batch of ``2**n_open`` amplitudes, which gives the pairwise steps a real
    N dimension.

    Returns ``(inputs, output, size_dict, arrays)`` with single-character index
    labels (``utils.get_symbol``), all of size 2.
    """
    if bitstring is None:
        bitstring = "0" * n
    counter = [0]

    def new_ix():
        ix = get_symbol(counter[0])
        counter[0] += 1
        return ix

    tensors = []  # [inds(list), array]
    cur = []
    zero, one = np.array([1.0, 0.0], dtype=complex), np.array([0.0, 1.0], dtype=complex)
    for q in range(n):
        ix = new_ix()
        cur.append(ix)
        tensors.append([[ix], zero.copy()])
    for name, qubits, params in gates:
        U = gate_matrix(name, params)
        if len(qubits) == 1:
            (q,) = qubits
            out = new_ix()
            tensors.append([[out, cur[q]], U.astype(complex)])
            cur[q] = out
        else:
            q0, q1 = qubits
            o0, o1 = new_ix(), new_ix()
            tensors.append([[o0, o1, cur[q0], cur[q1]], U.reshape(2, 2, 2, 2).astype(complex)])
            cur[q0], cur[q1] = o0, o1
    open_inds = []
    for q in range(n):
        if bitstring[q] in "01":
            tensors.append([[cur[q]], (one if bitstring[q] == "1" else zero).copy()])
        else:
            open_inds.append(cur[q])

    if simplify:
        tensors = absorb_low_rank(tensors, keep=set(open_inds))

    # relabel compactly in order of appearance
    relabel = {}
    inputs, arrays = [], []
    for inds, arr in tensors:
        term = []
        for ix in inds:
            if ix not in relabel:
                relabel[ix] = get_symbol(len(relabel))
            term.append(relabel[ix])
        inputs.append(tuple(term))
        arrays.append(np.ascontiguousarray(arr.astype(dtype)))
    size_dict = {ix: 2 for ix in relabel.values()}
    return inputs, tuple(relabel[ix] for ix in open_inds), size_dict, arrays


def absorb_low_rank(tensors, keep=()):
    """Contract every tensor of rank <= 2 into a neighbour (never raising the
    neighbour's rank), until none is left.  Indices in ``keep`` (open outputs)
    are never summed."""
    keep = set(keep)
    tensors = [[list(i), a] for i, a in tensors]
    where = {}
    for t, (inds, _) in enumerate(tensors):
        for ix in inds:
            where.setdefault(ix, set()).add(t)
    alive = set(range(len(tensors)))
    changed = True
    while changed:
        changed = False
        for t in sorted(alive):
            if t not in alive:
                continue
            inds, arr = tensors[t]
            if len(inds) > 2 or len(alive) == 1:
                continue
            # neighbour sharing an index, highest rank first
            nbrs = {u for ix in inds if ix not in keep for u in where[ix] if u != t and u in alive}
            if not nbrs:
                continue
            u = max(nbrs, key=lambda v: (len(tensors[v][0]), -v))
            uinds, uarr = tensors[u]
            shared = [ix for ix in inds if ix in uinds and ix not in keep]
            keep_t = [ix for ix in inds if ix not in shared]
            if len(keep_t) > 1:
                continue   # (open index + dangling index: leave the tensor alone)
            # result keeps u's index order with shared slots replaced by t's free index
            letters = {ix: chr(ord("a") + k) for k, ix in enumerate(dict.fromkeys(uinds + inds))}
            out = []
            replaced = False
            for ix in uinds:
                if ix in shared:
                    if keep_t and not replaced:
                        out.append(keep_t[0])
                        replaced = True
                else:
                    out.append(ix)
            if keep_t and not replaced:
                out.append(keep_t[0])
            eq = (
                "".join(letters[i] for i in uinds) + "," + "".join(letters[i] for i in inds)
                + "->" + "".join(letters[i] for i in out)
            )
            new = np.einsum(eq, uarr, arr)
            for ix in inds:
                where[ix].discard(t)
            for ix in shared:
                where[ix].discard(u)
            for ix in out:
                where.setdefault(ix, set()).add(u)
            tensors[u] = [out, new]
            alive.discard(t)
            changed = True
    return [tensors[t] for t in sorted(alive)]
