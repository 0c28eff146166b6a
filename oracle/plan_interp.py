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


def group_key(plan, slice_id):
    """``slice_id`` with the digits of the plan's group indices (``plan.slice_group``) set to zero: slices
    with the same key differ only in those and share every step marked ``group`` (the executor's
    ``slice_group_key``, csrc/ctg_runtime.hip)."""
    n_sl = len(plan.slice_sizes)
    flags = list(plan.slice_group) if len(plan.slice_group) == n_sl else [0] * n_sl
    key, rem, stride = 0, slice_id, 1
    for j in range(n_sl - 1, -1, -1):
        if plan.slice_fixed[j] >= 0:
            continue
        d = rem % plan.slice_sizes[j]
        rem //= plan.slice_sizes[j]
        if not flags[j]:
            key += d * stride
        stride *= plan.slice_sizes[j]
    return key


def group_members(plan, slice_id):
    """All slice ids that share ``slice_id``'s group (every value of the group indices)."""
    n_sl = len(plan.slice_sizes)
    flags = list(plan.slice_group) if len(plan.slice_group) == n_sl else [0] * n_sl
    strides, stride = [0] * n_sl, 1
    for j in range(n_sl - 1, -1, -1):
        if plan.slice_fixed[j] >= 0:
            continue
        strides[j] = stride
        stride *= plan.slice_sizes[j]
    ids = [group_key(plan, slice_id)]
    for j in range(n_sl):
        if flags[j] and plan.slice_fixed[j] < 0:
            ids = [i + d * strides[j] for i in ids for d in range(plan.slice_sizes[j])]
    return sorted(ids)


def run_stem2(step, spaces, base, chunk=256):
    """A fused stem pair (cotengra_amd/stem.py, csrc/ctg_stem.hip) executed from
    the very tables the kernel reads, tile by tile:

    * a tile's rows of A: grid offset + row-tile offset + chunk offset + slot
      offset + the constant of the lane that loads the element;
    * first product with B1 (``b1_off[k * N1 + n]``);
    * the intermediate tile laid out at ``mid_row[row] + mid_col[n]`` =
      ``row2 * ld2 + k2``;
    * second product with B2, stored at grid + ``out_row[row2] + out_col[n2]``.
    """
    st, T = step.stem, step.stem["tabs"]
    one = bool(st.get("one"))   # the first half alone: the product goes straight to the result
    tri = bool(st.get("KM"))    # a middle stage between the two (three-step tile)
    A, B1, C = (spaces[t.space] for t in (step.a, step.b, step.c))
    B2 = None if one else spaces[step.b2.space]
    K1, N1, K2, N2, ld2, rows2 = st["K1"], st["N1"], st["K2"], st["N2"], st["ld2"], st["rows2"]
    rows1 = 1 << st["nr1"]
    # element (row r, k) of a tile: row tile + chunk of 16 k + slot + the constant of the
    # lane that loads it (row r & 31, k-row h); k-row and slot of k by the gather mode
    r = np.arange(rows1)
    k = np.arange(K1)
    assert len(T["kj_a"]) == 8 and len(T["lane_a"]) == 64 and 8 * int(T["lane_a"].max()) < 1 << 32
    if st["vec"]:
        h, slot = (k >> 1) & 1, (((k & 15) >> 2) << 1) | (k & 1)
        assert np.all(T["kj_a"][1::2] == T["kj_a"][0::2] + 1)
    else:
        h, slot = k & 1, (k & 15) >> 1
    in_tile = (T["rt_a"][r >> 5][:, None] + T["chunk_a"][k >> 4][None, :] + T["kj_a"][slot][None, :]
               + T["lane_a"][(r & 31)[:, None] + 32 * h[None, :]])
    b1 = B1[base(step.b) + T["b1_off"]].reshape(K1, N1)
    out_at = T["out_row"][:, None] + T["out_col"][None, :]
    if one:
        assert out_at.shape == (rows1, N1) and len(np.unique(out_at)) == rows1 * N1
    else:
        b2 = B2[base(step.b2) + T["b2_off"]].reshape(K2, N2)
        mid_at = (T["mid_row"][:, None] + T["mid_col"][None, :]).reshape(-1)
        take = (np.arange(rows2)[:, None] * ld2 + np.arange(K2)[None, :])
        if tri:
            KM, NM, rows_m, ld_m = st["KM"], st["NM"], st["rowsM"], st["ldM"]
            bm = spaces[step.bm.space][base(step.bm) + T["bm_off"]].reshape(KM, NM)
            assert len(np.unique(mid_at)) == rows1 * N1 and mid_at.max() < rows_m * ld_m
            assert rows1 * N1 == rows_m * KM and rows_m * NM == rows2 * K2
            take_m = (np.arange(rows_m)[:, None] * ld_m + np.arange(KM)[None, :])
            mid2_at = (T["mid2_row"][:, None] + T["mid2_col"][None, :]).reshape(-1)
            assert len(np.unique(mid2_at)) == rows_m * NM and mid2_at.max() < rows2 * ld2
        else:
            assert len(np.unique(mid_at)) == rows1 * N1 and mid_at.max() < rows2 * ld2
            assert rows1 * N1 == rows2 * K2
    g_lo = st["g_lo"]
    for g0 in range(0, st["n_tiles"], chunk):
        g = np.arange(g0, min(g0 + chunk, st["n_tiles"]))
        ga = base(step.a) + T["gA_hi"][g // g_lo] + T["gA_lo"][g % g_lo]
        gc = base(step.c) + T["gC_hi"][g // g_lo] + T["gC_lo"][g % g_lo]
        a = A[ga[:, None, None] + in_tile[None]]                  # (g, rows1, K1)
        c1 = a @ b1                                               # (g, rows1, N1)
        if one:
            C[gc[:, None, None] + out_at[None]] = c1
            continue
        if tri:
            # three-step tile: the first intermediate [rowsM][ldM] x BM, its result laid out as the
            # second intermediate [rows2][ld2] through mid2_row[rowM] + mid2_col[nM]
            mid = np.zeros((len(g), rows_m * ld_m), dtype=c1.dtype)
            mid[:, mid_at] = c1.reshape(len(g), -1)
            cm = mid[:, take_m] @ bm                                  # (g, rowsM, NM)
            mid = np.zeros((len(g), rows2 * ld2), dtype=c1.dtype)
            mid[:, mid2_at] = cm.reshape(len(g), -1)
        else:
            mid = np.zeros((len(g), rows2 * ld2), dtype=c1.dtype)
            mid[:, mid_at] = c1.reshape(len(g), -1)
        a2 = mid[:, take]                                         # (g, rows2, K2)
        C[gc[:, None, None] + out_at[None]] = a2 @ b2


def run_lds_component(run, spaces, base, dt):
    """One LDS-resident subtree (``plan.lds_runs[i]``, cotengra_amd/ldsrun.py) as its workgroup runs
    it (csrc/ctg_lds_run.hip): the shadow records in order on a private array standing in for the LDS."""
    from cotengra_amd import ldsrun as L

    lds = np.full(max(int(run["lds_elems"]), 1), np.nan, dtype=dt)
    where = lambda t: lds if t.space == L.SPACE_LDS else spaces[t.space]  # noqa: E731
    at = lambda t: t.offset if t.space == L.SPACE_LDS else base(t)  # noqa: E731
    for sh in run["steps"]:
        st = sh["step"]
        if sh["kind"] == L.KIND_LOAD:
            ia = at(st.a) + _rows(st, "A")[:, None] + _ks(st, "A")[None, :]
            where(st.c)[at(st.c) + _rows(st, "C")] = where(st.a)[ia].sum(axis=1)
        else:
            rA, rB, rC = (_rows(st, k) for k in "ABC")
            kA, kB = _ks(st, "A"), _ks(st, "B")
            nB, nC = st.n_tabs["B"], st.n_tabs["C"]
            ia = at(st.a) + rA[:, None] + kA[None, :]
            ib = at(st.b) + rB[:, None, None] + kB[None, :, None] + nB[None, None, :]
            ic = at(st.c) + rC[:, None] + nC[None, :]
            where(st.c)[ic] = np.einsum("rk,rkn->rn", where(st.a)[ia], where(st.b)[ib])


def run_plan(plan, arrays, slice_ids=None, result=None, lds=False):
    """Execute ``plan`` for the given slices, accumulating into ``result``
    (a flat array of ``plan.result_elems``); returns the result reshaped.
    ``lds=True``: members of LDS-resident subtrees run through their shadow records
    (``plan.lds_runs``), all components of a sharing class at the first member -- the executor's order."""
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
    # (slice groups: same-key slices one after the other, the shared steps once per key -- the executor's order)
    grouped = any(getattr(st, "group", False) for st in plan.steps)
    if grouped:
        slice_ids = sorted(slice_ids, key=lambda i: (group_key(plan, i), i))
    last_key = None
    for sid in slice_ids:
        soff = slice_offsets(plan, sid)
        key = group_key(plan, sid) if grouped else None
        fresh, last_key = key != last_key, key

        def base(t):
            b = t.offset
            if t.leaf >= 0:
                b += int(soff[t.leaf])
            return b

        lds_done = set()
        for step in plan.steps:
            if step.invariant and not first:
                continue  # computed once, output persistent (like the executor)
            if grouped and getattr(step, "group", False) and not fresh:
                continue  # computed for the first slice of this group, what is read of it kept
            if lds and getattr(step, "lds_comp", -1) >= 0:
                cls = plan.lds_runs[step.lds_comp]["cls"]
                # (at the first member PAIR of the class: behind every step of the classes that run less often)
                if cls not in lds_done and step.kind == P.KIND_PAIR:
                    lds_done.add(cls)
                    for run in plan.lds_runs:
                        if run["cls"] == cls:
                            run_lds_component(run, spaces, base, dt)
                continue
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
            elif step.kind == P.KIND_STEM2:
                run_stem2(step, spaces, base)
            else:
                raise ValueError(f"bad step kind {step.kind}")
        first = False
    return result.reshape(plan.result_shape)
