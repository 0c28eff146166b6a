"""ORACLE -- test infrastructure, NOT product code.

A numpy interpreter for the *device plan* produced by
``cotengra_amd.plan.compile_tree``.  It applies exactly the addressing
semantics the HIP kernels implement (per-group offset tables, two-level row
tables, slice base offsets, arena offsets) so that the host planner can be
validated on a machine without a GPU: ``-m "not gpu"`` tests run a plan
through this interpreter and compare with ``oracle/contract_ref.py`` and the
golden vectors.  It is deliberately naive (gathers whole operands with fancy
indexing) and only suitable for small cases.
"""

from __future__ import annotations

import numpy as np

from cotengra_amd import plan as P


def _rows(step, key):
    hi, lo = step.rows[key]
    return (hi[:, None] + lo[None, :]).reshape(-1)


def _ks(step, key):
    hi, lo = step.k_tabs[key]
    return (hi[:, None] + lo[None, :]).reshape(-1)


def slice_offsets(plan, slice_id):
    """Base offset per input (and pseudo-input N = result chunk) for a slice:
    mixed-radix decode of ``slice_id`` over the sliced indices, most
    significant first, projected indices contributing their fixed value."""
    n_sl = len(plan.slice_sizes)
    digits = [0] * n_sl
    rem = slice_id
    for j in range(n_sl - 1, -1, -1):
        if plan.slice_fixed[j] >= 0:
            digits[j] = plan.slice_fixed[j]
        else:
            digits[j] = rem % plan.slice_sizes[j]
            rem //= plan.slice_sizes[j]
    if n_sl == 0:
        return np.zeros(len(plan.input_sizes) + 1, dtype=np.int64)
    return plan.slice_strides @ np.asarray(digits, dtype=np.int64)


def run_plan(plan, arrays, slice_ids=None, result=None):
    """Execute ``plan`` for the given slices, accumulating into ``result``
    (a flat array of ``plan.result_elems``); returns the result reshaped."""
    dt = np.dtype(plan.dtype)
    inputs = np.zeros(plan.inputs_elems, dtype=dt)
    for off, n, x in zip(plan.input_offsets, plan.input_sizes, arrays):
        inputs[off : off + n] = np.asarray(x, dtype=dt).reshape(-1)
    arena = np.zeros(plan.arena_elems, dtype=dt)
    if result is None:
        result = np.zeros(plan.result_elems, dtype=dt)
    spaces = {P.SPACE_INPUTS: inputs, P.SPACE_ARENA: arena, P.SPACE_RESULT: result}
    if slice_ids is None:
        slice_ids = range(plan.nslices)

    first = True
    for sid in slice_ids:
        soff = slice_offsets(plan, sid)

        def base(t):
            b = t.offset
            if t.leaf >= 0:
                b += int(soff[t.leaf])
            return b

        for step in plan.steps:
            if step.invariant and not first:
                continue  # computed once, output persistent (like the executor)
            if step.kind == P.KIND_SINGLE:
                src, dst = spaces[step.a.space], spaces[step.c.space]
                ia = base(step.a) + _rows(step, "A")[:, None] + _ks(step, "A")[None, :]
                ic = base(step.c) + _rows(step, "C")
                dst[ic] = src[ia].sum(axis=1)
            elif step.kind == P.KIND_ACCUM:
                src, dst = spaces[step.a.space], spaces[step.c.space]
                ia = base(step.a) + _rows(step, "A")
                ic = base(step.c) + _rows(step, "C")
                dst[ic] += src[ia]
            elif step.kind == P.KIND_PAIR:
                A, B, C = (spaces[t.space] for t in (step.a, step.b, step.c))
                kA, kB = _ks(step, "A"), _ks(step, "B")
                nB, nC = step.n_tabs["B"], step.n_tabs["C"]
                if step.kernel == P.KERNEL_MFMA:
                    rA, rC = _rows(step, "A"), _rows(step, "C")
                    for b in range(step.Bt):
                        ia = base(step.a) + step.b_tabs["A"][b] + rA[:, None] + kA[None, :]
                        ib = base(step.b) + step.b_tabs["B"][b] + kB[:, None] + nB[None, :]
                        ic = base(step.c) + step.b_tabs["C"][b] + rC[:, None] + nC[None, :]
                        C[ic] = A[ia] @ B[ib]
                else:
                    rA, rB, rC = (_rows(step, k) for k in "ABC")
                    ia = base(step.a) + rA[:, None] + kA[None, :]  # (R, K)
                    ib = base(step.b) + rB[:, None, None] + kB[None, :, None] + nB[None, None, :]
                    ic = base(step.c) + rC[:, None] + nC[None, :]
                    C[ic] = np.einsum("rk,rkn->rn", A[ia], B[ib])
            else:
                raise ValueError(f"bad step kind {step.kind}")
        first = False
    return result.reshape(plan.result_shape)
