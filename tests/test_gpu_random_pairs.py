"""GPU suite: randomised pairwise contractions through the C ABI.

Every kernel family has host-checked entry conditions (full tiles, additive
tables, 16-byte pairs, short K, streaming, k-streaming ...).  This sweep draws
index structures, extents and memory orders at random -- power-of-two networks
that land on the fast paths and ragged ones that must fall back -- and checks
each result against ``numpy.einsum`` in double precision.  Deterministic seeds.
"""
import numpy as np
import pytest

import cotengra_amd as ca
from cotengra_amd.interface import einsum

pytestmark = pytest.mark.gpu

LETTERS = "abcdefghijklmnopqrstuvwxyz"


def random_case(seed):
    rng = np.random.default_rng(seed)
    style = seed % 6
    # 0: power-of-two bits (Sycamore-like), 1: ragged dims, 2: tall-skinny pow2,
    # 3: long contraction / tiny result, 4: mixed with batch index, 5: small odd things
    if style in (0, 2, 3):
        pool = [2, 2, 2, 4]
    elif style == 1:
        pool = [2, 3, 5, 6, 7]
    else:
        pool = [2, 3, 4, 8]
    n_bat = int(rng.integers(0, 2)) if style in (4, 5) else 0
    budget = {0: 21, 1: 18, 2: 22, 3: 22, 4: 18, 5: 12}[style]

    def draw(lo, hi):
        return [int(rng.choice(pool)) for _ in range(int(rng.integers(lo, hi + 1)))]

    if style == 2:      # many rows, few k / n
        keep_a, con, keep_b = draw(12, 16), draw(1, 3), draw(0, 3)
    elif style == 3:    # long k, tiny result
        keep_a, con, keep_b = draw(0, 4), draw(14, 17), draw(0, 4)
    else:
        keep_a, con, keep_b = draw(1, 8), draw(0, 7), draw(0, 6)
    bat = draw(n_bat, n_bat)
    dims = bat + keep_a + con + keep_b
    names = list(LETTERS[: len(dims)])
    size = dict(zip(names, dims))
    nb, na, nc = len(bat), len(keep_a), len(con)
    ib, ia, ic, ik = names[:nb], names[nb:nb + na], names[nb + na:nb + na + nc], names[nb + na + nc:]

    def log2(ix):
        return sum(np.log2(size[i]) for i in ix)

    # stay inside the budget (log2 elements of the largest operand / the MAC count)
    while log2(ib + ia + ic) > budget or log2(ib + ic + ik) > budget or log2(ib + ia + ik) > budget \
            or log2(ib + ia + ic + ik) > 31:
        for grp in (ia, ic, ik):
            if len(grp) > 1 and (log2(ib + ia + ic) > budget or log2(ib + ic + ik) > budget
                                 or log2(ib + ia + ik) > budget or log2(ib + ia + ic + ik) > 31):
                grp.pop(int(rng.integers(0, len(grp))))
        if len(ia) <= 1 and len(ic) <= 1 and len(ik) <= 1:
            break
    ta = list(rng.permutation(ib + ia + ic))
    tb = list(rng.permutation(ib + ic + ik))
    to = list(rng.permutation(ib + ia + ik))
    if not ta or not tb:
        return None
    eq = "".join(ta) + "," + "".join(tb) + "->" + "".join(to)
    return eq, size


CASES = [c for c in (random_case(s) for s in range(240)) if c is not None]


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_random_pair(idx):
    eq, size = CASES[idx]
    dtype = ["complex64", "complex64", "complex64", "complex128", "float32", "float64"][idx % 6]
    (ta, tb), _ = ca.eq_to_inputs_output(eq)
    rng = np.random.default_rng(1000 + idx)

    def mk(t):
        shape = [size[i] for i in t]
        x = rng.normal(size=shape)
        if "complex" in dtype:
            x = x + 1j * rng.normal(size=shape)
        return x.astype(dtype)

    a, b = mk(ta), mk(tb)
    hi = "complex128" if "complex" in dtype else "float64"
    ref = np.einsum(eq, a.astype(hi), b.astype(hi), optimize=True)
    got = np.asarray(einsum(eq, a, b, optimize=[(0, 1)]))
    assert got.shape == ref.shape
    scale = max(np.abs(ref).max(), 1e-300)
    tol = 5e-4 if dtype in ("complex64", "float32") else 1e-11
    assert np.abs(got - ref).max() <= tol * scale, (eq, size, dtype)
